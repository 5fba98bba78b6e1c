// Which XCD does workgroup L of a launch run on?  Prints the XCC id (HW_REG_XCC_ID) of every workgroup for a few grid sizes,
// twice per size (is the assignment the same for consecutive launches?).  hipcc --offload-arch=gfx950 xcc_map.hip -o xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void who(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * blockIdx.y] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;
}

int main() {
    int* d;
    hipMalloc(&d, 4096 * sizeof(int));
    const int gx[] = {256, 80, 128, 32}, gy[] = {1, 1, 1, 4};
    for (int c = 0; c < 4; ++c)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0xff, 4096 * sizeof(int));
            hipLaunchKernelGGL(who, dim3(gx[c], gy[c]), dim3(256), 0, 0, d);
            hipDeviceSynchronize();
            std::vector<int> h(gx[c] * gy[c]);
            hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
            int rr = 0;
            for (size_t i = 0; i < h.size(); ++i) rr += (h[i] == (int)(i % 8));
            printf("grid (%d,%d) launch %d: %d of %zu workgroups on XCD id %% 8; first 24:", gx[c], gy[c], rep, rr, h.size());
            for (int i = 0; i < 24 && i < (int)h.size(); ++i) printf(" %d", h[i]);
            printf("\n");
        }
    return 0;
}
