// Experiment: what does a device-wide barrier inside one resident grid cost on MI355X (256 CUs, 8 XCDs with private L2s), against the
// ~5.3 us a dependent kernel launch costs inside a hipGraph (tools/probes/launch_floor.py)?  It decides whether the denoise loop's
// 1080 tiny dependent launches could become phases of one persistent kernel.
//   build: hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o tools/probes/grid_barrier.bin
// Each iteration: every block writes a value, barrier, every block reads another block's value (on another XCD) and checks it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) break;  // never hang the box
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int WORK>
__global__ __launch_bounds__(512) void probe(unsigned* ctr, int iters, float* data, int stride, unsigned* bad, const float* big, float* sink) {
    const int nb = gridDim.x;
    unsigned wrong = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (WORK) {  // stream 64 KiB per block from a 16 MiB+ buffer (a phase's weights)
            const float4* src = reinterpret_cast<const float4*>(big) + ((size_t)((it * nb + blockIdx.x) & 1023) * 4096);
            for (int i = threadIdx.x; i < 4096; i += 512) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
        }
        for (int i = threadIdx.x; i < stride; i += 512) data[(size_t)blockIdx.x * stride + i] = (float)(it * 1000 + blockIdx.x);
        grid_barrier(ctr, (unsigned)(it + 1) * nb);
        const int other = (blockIdx.x + 97) % nb;  // 97 is odd: another XCD
        for (int i = threadIdx.x; i < stride; i += 512)
            wrong += data[(size_t)other * stride + i] != (float)(it * 1000 + other);
        grid_barrier(ctr + 32, (unsigned)(it + 1) * nb);  // (WAR: nobody overwrites before everybody has read)
    }
    if (wrong) atomicAdd(bad, wrong);
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ void empty_kernel(float* p) { if (p == nullptr) return; }

int main() {
    unsigned *ctr, *bad; float *data, *big, *sink;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&data, 256 * 4096 * 4)); CK(hipMalloc(&big, (size_t)1024 * 65536)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(big, 0, (size_t)1024 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nb : {64, 128, 256}) for (int stride : {1, 1024}) for (int work = 0; work < 2; ++work) {
        float best = 1e9f; unsigned hb = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0));
            if (work) hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(512), 0, 0, ctr, iters, data, stride, bad, big, sink);
            else hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(512), 0, 0, ctr, iters, data, stride, bad, big, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        }
        printf("blocks %3d  %4d floats per block exchanged  %s: %.2f us per iteration (two barriers), stale reads %u\n", nb, stride,
               work ? "64 KiB streamed per block" : "no other work            ", best * 1e3f / iters, hb);
    }
    // the alternative: dependent launches of an empty kernel, captured in a graph
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, s, data);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1000 dependent empty launches (256 x 512) in a graph: %.2f us per launch\n", ms);
    return 0;
}
