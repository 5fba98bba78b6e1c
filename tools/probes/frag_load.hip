// How fast can one CU pull MFMA fragments straight from global memory?  The skinny (M <= 64 rows) GEMMs of the denoise loop read
// their operands as 16 rows x 64 B per wave-instruction (lane (i, g): row i, bytes 16 g .. 16 g + 15 of a 64-B run).  Compare
// with the same bytes as ONE contiguous 1-KiB run per wave-instruction (a fragment-major packed layout).
//   build: hipcc --offload-arch=gfx950 -O3 tools/probes/frag_load.hip -o /tmp/frag_load && /tmp/frag_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// each wave: NL loads of 16 B per lane.  mode 0: rows of `ld` bytes, lane (i = lane & 15, g = lane >> 4) reads row i, 64-B run j
// mode 1: load j reads bytes [1024 j, 1024 j + 1024) of the wave's private contiguous region
template <int NL, int MODE, bool SHARED>
__global__ __launch_bounds__(1024) void frag_load(const char* __restrict__ base, int64_t per_block, int ld, uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* blk = base + (SHARED ? 0 : (int64_t)blockIdx.x * per_block);
    u32x4 v[NL];
    if constexpr (MODE == 0) {
        // the wave owns K bytes [wave * NL * 64, ...) of 16 rows: row stride ld
        const char* p = blk + (int64_t)(lane & 15) * ld + (int64_t)wave * NL * 64 + (lane >> 4) * 16;
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = *reinterpret_cast<const u32x4*>(p + j * 64);
    } else if constexpr (MODE >= 2) {
        // MODE = rows per instruction (8, 4, 2): R rows x (1024 / R) contiguous bytes; the wave owns K bytes [wave * NL * 1024 / R, ...) of R rows
        constexpr int R = MODE >= 2 ? MODE : 2, SEG = 1024 / R, LPR = SEG / 16;
        const char* p = blk + (int64_t)(lane / LPR) * ld + (int64_t)wave * NL * SEG + (lane % LPR) * 16;
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = *reinterpret_cast<const u32x4*>(p + j * SEG);
    } else {
        const char* p = blk + (int64_t)wave * NL * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < NL; ++j) v[j] = *reinterpret_cast<const u32x4*>(p + j * 1024);
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) s ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
    if (s == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NL, int MODE, bool SHARED>
float run(const char* buf, int64_t per_block, int ld, uint32_t* out, int waves, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((frag_load<NL, MODE, SHARED>), dim3(256), dim3(waves * 64), 0, 0, buf, per_block, ld, out);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((frag_load<NL, MODE, SHARED>), dim3(256), dim3(waves * 64), 0, 0, buf, per_block, ld, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters * 1e3f;
}

int main() {
    const int64_t per_block = 1 << 20;  // 1 MiB apart
    char* buf;
    uint32_t* out;
    hipMalloc(&buf, 256 * per_block + (1 << 20));
    hipMemset(buf, 1, 256 * per_block + (1 << 20));
    hipMalloc(&out, 256 * 1024 * 4);
    // bytes per block = waves * NL * 1 KiB
    printf("pattern                              KiB/CU   private(us)  shared(us)\n");
#define ROW(NL, W)                                                                                                              \
    printf("16 rows x 64 B  (%2d loads x %2d waves) %6d   %8.2f   %8.2f\n", NL, W, NL * W,                                        \
           run<NL, 0, false>(buf, per_block, NL * W * 64, out, W, 200), run<NL, 0, true>(buf, per_block, NL * W * 64, out, W, 200)); \
    printf("1 KiB contiguous (%2d loads x %2d waves) %6d   %8.2f   %8.2f\n", NL, W, NL * W,                                       \
           run<NL, 1, false>(buf, per_block, 0, out, W, 200), run<NL, 1, true>(buf, per_block, 0, out, W, 200));
#define ROWR(R, NL, W)                                                                                                          \
    printf("%d rows x %4d B  (%2d loads x %2d waves) %6d   %8.2f   %8.2f\n", R, 1024 / R, NL, W, NL * W,                          \
           run<NL, R, false>(buf, per_block, NL * W * (1024 / R), out, W, 200), run<NL, R, true>(buf, per_block, NL * W * (1024 / R), out, W, 200));
    ROWR(8, 16, 16)
    ROWR(4, 16, 16)
    ROWR(2, 16, 16)
    ROWR(8, 8, 16)
    ROWR(4, 8, 16)
    ROWR(2, 8, 16)
    ROW(8, 4)
    ROW(16, 4)
    ROW(8, 16)
    ROW(16, 16)
    ROW(32, 8)
    ROW(32, 16)
    // empty-ish kernel for the floor
    printf("floor (1 load x 1 wave): %.2f us\n", run<1, 1, true>(buf, per_block, 0, out, 1, 200));
    return 0;
}
