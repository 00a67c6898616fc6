#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int x = 1000 + lane, y = 2000 + lane;
    auto r32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[lane] = r32[0]; out[64 + lane] = r32[1];
    auto r16 = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[128 + lane] = r16[0]; out[192 + lane] = r16[1];
    int v = lane * 10;
    out[256 + lane] = __builtin_amdgcn_readlane(v, (lane * 0) + 37);
}
int main() {
    int* d; hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"p32 vdst'", "p32 src'", "p16 vdst'", "p16 src'"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%d", l, h[a * 64 + l]); printf("\n"); }
    printf("readlane37 = %d\n", h[256]);
    return 0;
}
