// Does the raw-buffer range check on gfx950 include the SGPR offset (soffset)?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/soffset_check.hip -o tools/ubench/soffset_check && tools/ubench/soffset_check
// A 1 KiB array filled with 7.0f; descriptor num_records = 16 bytes.  Loads:
//   (a) voffset = 64, soffset = 0   -> out of range by the vector offset: expect 0
//   (b) voffset = 0,  soffset = 64  -> the address is past num_records only through the SGPR offset
// If (b) returns 7 the check ignores soffset: row offsets of clipped tiles must then travel in the VECTOR offset.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 16, 0x00020000);
    const int soff = __builtin_amdgcn_readfirstlane(64);
    out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 64, 0, 0));
    out[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 0, soff, 0));
    out[2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 0));
}
int main() {
    float *x, *o, h[256], r[3];
    for (float& v : h) v = 7.0f;
    hipMalloc(&x, sizeof(h)); hipMalloc(&o, 12);
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o);
    hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
    printf("voffset past num_records: %g   soffset past num_records: %g   in range: %g\n", r[0], r[1], r[2]);
    printf("%s\n", r[1] == 0.0f ? "soffset IS range-checked" : "soffset is NOT range-checked");
    return 0;
}
