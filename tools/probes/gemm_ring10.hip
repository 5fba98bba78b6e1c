// Experiment: the NT 256x256x64 quadrant schedule on a 10-slot ring of 16 KiB half-tiles (all 160 KiB of LDS) instead of two
// 64 KiB stages, with the first B half's fragments kept in registers for the whole K-tile (so every half is read in ONE slot and
// its ring slot is free right after).  Every half-tile is then requested 8-10 slots (2+ K-tiles) before it is read, against 4-6 in
// the production schedule.  Question: is the K loop bound by how far ahead the LDS-DMA runs?
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gemm_ring10.hip -Lkai0_amd/lib -lkai0hip -Wl,-rpath,$PWD/kai0_amd/lib -o tools/probes/gemm_ring10.bin
//   run:   tools/probes/gemm_ring10.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
#include "../../include/kai0hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDS_PTR(T) __attribute__((address_space(3))) T*
constexpr uint32_t OOB = 0x80000000u;
constexpr int BK = 64, HALF = 16384, NSLOT = 10;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))lds_dst_wave_uniform, 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int VARIANT>
__global__ __launch_bounds__(512, 1) void gemm_ring10(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                      int M, int N, int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wm;
    const int l15 = lane & 15, g = lane >> 4;
    // block -> tile: XCD-aware bijective remap, grouped raster (the production kernel's)
    int pid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = pid & 7, idx = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP = 4;
    const int width = GROUP * tiles_n, group = pid / width, first_m = group * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP), in_g = pid - group * width;
    const int m0 = (first_m + in_g % gsz) * 256, n0 = (in_g / gsz) * 256;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)OOB, 0x00020000);
    const int kc_chunk = ((lane & 7) ^ (lane >> 3)) * 8;
    const int nk = K / BK;
    const uint32_t lda2 = (uint32_t)K * 2, ldb2 = (uint32_t)K * 2, kc2 = (uint32_t)kc_chunk * 2;
    // half-tile kinds in need order within a K-tile: 0 = B half 0, 1 = A half 0, 2 = B half 1, 3 = A half 1
    // A half h = rows wm*128 + h*64 + [0,64) of both wave rows (slot layout [wm][64 rows][128 B]); B half h = columns wn*64 + h*32 + [0,32)
    uint32_t src_off[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = 2 * wave + j;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int Ra = m0 + (q >> 3) * 128 + h * 64 + (q & 7) * 8 + (lane >> 3);
            const int Rb = n0 + (q >> 2) * 64 + h * 32 + (q & 3) * 8 + (lane >> 3);
            src_off[1 + 2 * h][j] = (uint32_t)Ra * lda2 + kc2;
            src_off[2 * h][j] = (uint32_t)Rb * ldb2 + kc2;
        }
    }
    const int dst_lds0 = (2 * wave) * 1024;  // piece j of this wave lands at slot + (2*wave + j) * 1 KiB
    auto issue = [&](auto kindc, int tt, int slot) {  // half `kind` of K-tile tt into ring slot `slot`
        constexpr int kind = decltype(kindc)::value;
        const bool in = tt < nk;
        char* base = smem + slot * HALF + dst_lds0;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            glds16((kind & 1) ? a_rsrc : b_rsrc, in ? src_off[kind][j] + (uint32_t)tt * (BK * 2) : OOB, base + j * 1024);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };
    // fragment addresses inside a slot: row r (128 B), 16-B chunk (ks*4 + g) ^ (r & 7)
    const int fa0 = (wm * 64 + l15) * 128 + ((g ^ (l15 & 7)) << 4);
    const int fb0 = (wn * 32 + l15) * 128 + ((g ^ (l15 & 7)) << 4);
    auto frag = [&](int slot, int lane_off, int tile16, int ks) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(smem + slot * HALF + ((lane_off + tile16 * 2048) ^ (ks << 6)));
    };

    // ---- prologue: halves 0..9 = K-tiles 0, 1 and the first two halves of K-tile 2 --------------------------------------------
    issue(K0{}, 0, 0); issue(K1{}, 0, 1); issue(K2{}, 0, 2); issue(K3{}, 0, 3);
    issue(K0{}, 1, 4); issue(K1{}, 1, 5); issue(K2{}, 1, 6); issue(K3{}, 1, 7);
    issue(K0{}, 2, 8); issue(K1{}, 2, 9);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // halves 0 and 1 of this wave have landed
    lds_barrier();
    if (grp == 1) lds_barrier();  // the two wave rows run one slot apart

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], b0[2][2], b1[2][2];
    auto quad = [&](auto ahc, auto bhc, bf16x8 (&bq)[2][2]) {
        constexpr int ah = decltype(ahc)::value, bh = decltype(bhc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[ah * 4 + i][bh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bq[j][ks], acc[ah * 4 + i][bh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    int s0 = 0;  // ring slot of K-tile t's first half (B half 0); the other three follow
    if (VARIANT == 2) {
        // one half-tile per load slot: R1 <- A half 0 of t+2, R2 <- B half 1 of t+2, R3 <- A half 1 of t+2, R4 <- B half 0 of t+3
        for (int t = 0; t < nk; ++t) {
            const int sB0 = s0, sA0 = wrap(s0 + 1), sB1 = wrap(s0 + 2), sA1 = wrap(s0 + 3), sPrev = wrap(s0 + 9);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b0[j][ks] = frag(sB0, fb0, j, ks);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[i][ks] = frag(sA0, fa0, i, ks);
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) issue(K1{}, t + 2, sPrev);
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // B half 1 of t
            lds_barrier();
            quad(K0{}, K0{}, b0);
            lds_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b1[j][ks] = frag(sB1, fb0, j, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(K2{}, t + 2, sB0);
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // A half 1 of t
            lds_barrier();
            quad(K0{}, K1{}, b1);
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[i][ks] = frag(sA1, fa0, i, ks);
            __builtin_amdgcn_sched_barrier(0);
            issue(K3{}, t + 2, sA0);
            lds_barrier();
            quad(K1{}, K1{}, b1);
            lds_barrier();
            issue(K0{}, t + 3, sB1);
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // B half 0 and A half 0 of t + 1
            lds_barrier();
            quad(K1{}, K0{}, b0);
            lds_barrier();
            s0 = wrap(s0 + 4);
        }
    } else
    for (int t = 0; t < nk; ++t) {
        const int sB0 = s0, sA0 = wrap(s0 + 1), sB1 = wrap(s0 + 2), sA1 = wrap(s0 + 3), sPrev = wrap(s0 + 9);
        // ---- slot R1: B half 0 and A half 0 into registers; the slot A half 1 of the previous K-tile left gets A half 0 of t + 2
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b0[j][ks] = frag(sB0, fb0, j, ks);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = frag(sA0, fa0, i, ks);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) issue(K1{}, t + 2, sPrev);
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // B half 1 of t (read two barriers from here) has landed
        lds_barrier();
        quad(K0{}, K0{}, b0);
        lds_barrier();
        // ---- slot R2: B half 1; the two slots read in R1 get B half 1 and A half 1 of t + 2
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) b1[j][ks] = frag(sB1, fb0, j, ks);
        __builtin_amdgcn_sched_barrier(0);
        issue(K2{}, t + 2, sB0);
        issue(K3{}, t + 2, sA0);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // A half 1 of t
        lds_barrier();
        quad(K0{}, K1{}, b1);
        lds_barrier();
        // ---- slot R3: A half 1; the slot read in R2 gets B half 0 of t + 3
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = frag(sA1, fa0, i, ks);
        __builtin_amdgcn_sched_barrier(0);
        issue(K0{}, t + 3, sB1);
        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // B half 0 and A half 0 of t + 1
        lds_barrier();
        if (VARIANT == 0) {
            quad(K1{}, K1{}, b1);
            lds_barrier();
            lds_barrier();  // (slot R4: nothing to read — B half 0 is still in registers)
            quad(K1{}, K0{}, b0);
            lds_barrier();
        } else {  // both remaining quadrants in one MFMA slot
            quad(K1{}, K1{}, b1);
            quad(K1{}, K0{}, b0);
            lds_barrier();
        }
        s0 = wrap(s0 + 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grp == 0) lds_barrier();
    lds_barrier();
    // ---- epilogue: accumulators -> wave-private f32 slab [16][64] -> 16-B bf16 stores ------------------------------------------
    float* slab = reinterpret_cast<float*>(smem + wave * 4096);
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(4 * g + r) * 64 + j * 16 + l15] = acc[ti][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // row-tile ti of half ah = ti / 4 holds tile rows wm*128 + ah*64 + (ti%4)*16 + [0,16); column tiles j of half bh = j / 2
        // hold columns wn*64 + bh*32 + (j%2)*16 + [0,16): the same row / column order as the accumulator indices
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
            const float* sp = slab + r * 64 + c8;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)sp[e];
            const int row = m0 + wm * 128 + ti * 16 + r, col = n0 + wn * 64 + c8;
            *reinterpret_cast<bf16x8*>(C + (int64_t)row * N + col) = o;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int shapes[][3] = {{30976, 2048, 16384}, {30976, 16384, 2048}, {30976, 2048, 2048}, {8192, 8192, 8192},
                             {24576, 4352, 1152}, {24576, 3584, 1152}, {24576, 1280, 1152}, {24576, 1280, 4352}, {30976, 2560, 2048}};  // (SigLIP-like: N, K padded to the probe's multiples)
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        auto gauss = [&]() { return (rnd() + rnd() + rnd() + rnd()) * 1.732f; };  // ~N(0, 1)
        for (auto& v : hA) v = f2bf(gauss());
        for (auto& v : hB) v = f2bf(gauss() * 0.03f);
        bf16_t *dA, *dB, *dC, *dR;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dR, (size_t)M * N * 2));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        const int tm = M / 256, tn = N / 256;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](auto fn) {
            for (int i = 0; i < 3; ++i) fn();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) fn();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12;
        };
        kai0_gemm_desc d; memset(&d, 0, sizeof d);
        d.A = dA; d.B = dB; d.C = dR; d.M = M; d.N = N; d.K = K; d.a_kc = 1; d.b_kc = 1; d.lda = K; d.ldb = K; d.ldc = N;
        d.batch = 1; d.batch_inner = 1; d.scale = 1.0f; d.split_k = 1;
        kai0_gemm_set_persist(0);
        const double t_plain = timeit([&] { if (kai0_gemm_bf16(&d, nullptr)) { printf("gemm: %s\n", kai0_last_error()); exit(1); } });
        kai0_gemm_set_persist(2);
        const double t_pers = timeit([&] { kai0_gemm_bf16(&d, nullptr); });
        kai0_gemm_set_persist(1);
        CK(hipFuncSetAttribute((const void*)gemm_ring10<0>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * HALF));
        CK(hipFuncSetAttribute((const void*)gemm_ring10<1>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * HALF));
        CK(hipFuncSetAttribute((const void*)gemm_ring10<2>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * HALF));
        const double t_r0 = timeit([&] { hipLaunchKernelGGL(gemm_ring10<0>, dim3(tm * tn), dim3(512), NSLOT * HALF, 0, dA, dB, dC, M, N, K, tm, tn); });
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        // compare with the production result
        std::vector<uint16_t> hC((size_t)M * N), hR((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hR.data(), dR, hR.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0; double maxd = 0;
        for (size_t i = 0; i < hC.size(); ++i) {
            if (hC[i] != hR[i]) { ++bad; maxd = fmax(maxd, fabs(bf2f(hC[i]) - bf2f(hR[i]))); }
        }
        const double t_r1 = timeit([&] { hipLaunchKernelGGL(gemm_ring10<1>, dim3(tm * tn), dim3(512), NSLOT * HALF, 0, dA, dB, dC, M, N, K, tm, tn); });
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        size_t bad1 = 0;
        for (size_t i = 0; i < hC.size(); ++i) bad1 += hC[i] != hR[i];
        const double t_r2 = timeit([&] { hipLaunchKernelGGL(gemm_ring10<2>, dim3(tm * tn), dim3(512), NSLOT * HALF, 0, dA, dB, dC, M, N, K, tm, tn); });
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        size_t bad2 = 0;
        for (size_t i = 0; i < hC.size(); ++i) bad2 += hC[i] != hR[i];
        printf("   balanced ring10 %7.1f (mismatches %zu)\n", t_r2, bad2);
        printf("%6d x %6d x %6d: production plain %7.1f  persistent %7.1f | ring10 %7.1f (mismatches %zu, max |d| %.3g)  ring10 3-slot tail %7.1f (mismatches %zu) TFLOP/s\n",
               M, N, K, t_plain, t_pers, t_r0, bad, maxd, t_r1, bad1);
        fflush(stdout);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dR);
    }
    return 0;
}
