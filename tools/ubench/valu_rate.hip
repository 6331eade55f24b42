// Micro-benchmark: cycles per wave64 VALU instruction on one SIMD for several opcodes, 8 waves per SIMD.
// Each kernel runs a long dependent-per-register but 8-way independent chain of one opcode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ITER 2000
#define DEFK(name, body)                                                                      \
    __global__ void __launch_bounds__(256) k_##name(uint32_t *out, uint32_t s) {              \
        uint32_t r[8];                                                                        \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7 + i + s;                           \
        uint32_t c = s | 1;                                                                   \
        for (int it = 0; it < N_ITER; ++it) {                                                 \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { body; }                            \
        }                                                                                     \
        uint32_t acc = 0;                                                                     \
        for (int i = 0; i < 8; ++i) acc ^= r[i];                                              \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                            \
    }
DEFK(add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c)))
DEFK(xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c)))
DEFK(alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(r[i])))
DEFK(add3, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)))
DEFK(perm, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)))
DEFK(xad, asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)))
DEFK(bfi, asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r[i]) : "v"(c)))
DEFK(cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c)))
DEFK(fmaf, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c)))
DEFK(addf, asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c)))
DEFK(lshl, asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i])))
DEFK(mul24, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(c)))
DEFK(mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c)))
DEFK(bcnt, asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c)))

template <typename K> void run(const char *name, K kern, uint32_t *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU = 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)blocks * 4 /*waves*/ * N_ITER * 8 / 1024.0;
    printf("%-10s %8.1f us  -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms * 1e3,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    uint32_t *out; hipMalloc(&out, 256 * 8 * 256 * 4);
#define R(n) run(#n, k_##n, out)
    R(add); R(add); R(xor); R(alignbit); R(add3); R(perm); R(xad); R(bfi); R(cndmask); R(fmaf); R(addf); R(lshl); R(mul24); R(mullo); R(bcnt);
    return 0;
}
