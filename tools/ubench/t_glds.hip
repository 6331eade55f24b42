#include <hip/hip_runtime.h>
__global__ void k(const uint4* src, uint4* out) {
    __shared__ __attribute__((aligned(16))) char buf[4096];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // each wave fills 1 KiB at buf + 1024*w; lane i's 16 bytes land at +16*i; source chosen per lane (reversed here)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w * 64 + (63 - lane)),
                                     (__attribute__((address_space(3))) void*)(buf + 1024 * w), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[threadIdx.x] = reinterpret_cast<uint4*>(buf)[threadIdx.x];
}
int main() {
    uint4 *s, *o; hipMalloc(&s, 256 * 16); hipMalloc(&o, 256 * 16);
    uint4 h[256]; for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i, i, i);
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s, o);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 256; ++i) { int w = i / 64, l = i % 64; if (h[i].x != (unsigned)(w * 64 + 63 - l)) ok = 0; }
    printf("global_load_lds dwordx4: %s (h[0]=%u h[1]=%u h[64]=%u)\n", ok ? "OK" : "MISMATCH", h[0].x, h[1].x, h[64].x);
    return 0;
}
