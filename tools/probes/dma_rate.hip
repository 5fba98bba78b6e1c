// Per-CU staging rate into LDS: LDS-DMA (buffer_load ... lds, 16 B per lane = 1 KiB per wave-instruction, 8 rows x 128 B — the
// pattern of the bf16 GEMM's K-contiguous tiles) against the same bytes loaded into VGPRs and written with ds_write_b128.
//   build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_rate.hip -o tools/probes/dma_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define LDS_PTR(T) __attribute__((address_space(3))) T*

// block = W waves; each wave moves NL KiB per round, ROUNDS rounds (double-buffer style: wait for all of a round, then next)
template <int NL, int MODE, bool SHARED>
__global__ __launch_bounds__(512) void stage(const char* __restrict__ base, int64_t per_block, int rounds, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* blk = base + (SHARED ? 0 : (int64_t)blockIdx.x * per_block);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)blk, 0, 0x7fffffff, 0x00020000);
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        // 8 rows x 128 B per instruction: row = lane >> 3 (row stride 8 KiB), 16-B slot lane & 7; piece j -> next 8 rows
        const uint32_t off0 = (uint32_t)(((r * nw + wave) * NL) * 8 * 8192 + (lane >> 3) * 8192 + (lane & 7) * 16) % (2u << 20);
        char* dst = smem + ((r & 1) * nw + wave) * NL * 1024;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < NL; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_PTR(void))(dst + j * 1024), 16, (int)((off0 + j * 8 * 8192) % (2u << 20)), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) v[j] = *reinterpret_cast<const u32x4*>(blk + (off0 + j * 8 * 8192) % (2u << 20));
#pragma unroll
            for (int j = 0; j < NL; ++j) *reinterpret_cast<u32x4*>(dst + j * 1024 + lane * 16) = v[j];
        }
        __syncthreads();
        acc ^= *reinterpret_cast<const uint32_t*>(smem + ((r & 1) * nw + wave) * NL * 1024 + lane * 4);
    }
    if (acc == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NL, int MODE, bool SHARED>
float run(const char* buf, int64_t per_block, int rounds, uint32_t* out, int waves) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int lds = 2 * waves * NL * 1024;
    hipFuncSetAttribute((const void*)stage<NL, MODE, SHARED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stage<NL, MODE, SHARED>), dim3(256), dim3(waves * 64), lds, 0, buf, per_block, rounds, out);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stage<NL, MODE, SHARED>), dim3(256), dim3(waves * 64), lds, 0, buf, per_block, rounds, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 20 * 1e3f;
}

int main() {
    const int64_t per_block = 2 << 20;
    char* buf;
    uint32_t* out;
    hipMalloc(&buf, 256 * per_block + (64 << 20));
    hipMemset(buf, 1, 256 * per_block + (64 << 20));
    hipMalloc(&out, 256 * 512 * 4);
    const int rounds = 32;
    printf("8 waves x NL KiB per round, %d rounds (a K loop of %d tiles); GB/s per CU\n", rounds, rounds);
#define ROW(NL)                                                                                                     \
    {                                                                                                               \
        const double kb = 8.0 * NL * rounds;                                                                        \
        const float d0 = run<NL, 0, false>(buf, per_block, rounds, out, 8), d1 = run<NL, 0, true>(buf, per_block, rounds, out, 8); \
        const float v0 = run<NL, 1, false>(buf, per_block, rounds, out, 8), v1 = run<NL, 1, true>(buf, per_block, rounds, out, 8); \
        printf("%2d KiB/round/CU: LDS-DMA private %6.1f us (%5.1f GB/s) shared %6.1f us (%5.1f) | VGPR+ds_write private %6.1f us (%5.1f) shared %6.1f us (%5.1f)\n", \
               8 * NL, d0, kb * 1.024e-3 / d0 * 1e3 / 1e3 * 1e3 / 1e3, d1, kb * 1.024 / d1, v0, kb * 1.024 / v0, v1, kb * 1.024 / v1);              \
    }
    ROW(2)
    ROW(4)
    ROW(8)
    return 0;
}
