// MFMA building blocks shared by the fused update kernels (csrc/g2048_tail.hip, csrc/g2048_block.hip): fragment layouts of
// v_mfma_f32_32x32x16_bf16, the fragment-packed weight layout, the register ring that streams weight units ahead of their MFMAs,
// LDS activation tiles, the update's dropout hash.  Device code only; included inside the including file's anonymous namespace
// users via `using namespace g2048_mfma`.
#ifndef G2048_MFMA_H
#define G2048_MFMA_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace g2048_mfma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// accumulator register i of lane (r, h) holds row rowof(i, h), column r of a 32 x 32 tile
__device__ __forceinline__ int rowof(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// the update's dropout hash (same function as csrc/g2048_layernorm.hip)
__device__ __forceinline__ bool keep_elem(uint32_t s0, uint32_t s1, uint32_t thr, uint64_t idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ s0;
    x ^= (uint32_t)(idx >> 32) * 0x85EBCA77u + s1;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (x >> 8) >= thr;
}
struct Drop {
    uint32_t s0, s1, thr;
    float inv_keep;
    __device__ __forceinline__ Drop site(uint32_t k) const { return Drop{s0 + k * 0x632BE5ABu, s1 ^ (k * 0x7F4A7C15u), thr, inv_keep}; }
    // four consecutive elements idx .. idx + 3 (idx a multiple of 4): one hash per PAIR, its two 16-bit halves compared with the
    // threshold at 16-bit resolution (the convention of g2048_relu_dropout_fwd): half the vector instructions of four full hashes
    __device__ __forceinline__ void apply4(float v[4], uint64_t idx) const {
        if (!thr) return;
        const uint32_t thr16 = thr >> 8;
        for (int pr = 0; pr < 2; ++pr) {
            const uint64_t id = (idx >> 1) + pr;
            uint32_t x = (uint32_t)id * 0x9E3779B1u ^ s0;
            x ^= (uint32_t)(id >> 32) * 0x85EBCA77u + s1;
            x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
            v[2 * pr] = (x & 0xFFFFu) >= thr16 ? v[2 * pr] * inv_keep : 0.0f;
            v[2 * pr + 1] = (x >> 16) >= thr16 ? v[2 * pr + 1] * inv_keep : 0.0f;
        }
    }
};
__device__ __forceinline__ Drop make_drop(uint64_t seed, const uint64_t *seed_state, float p_drop) {
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    if (seed_state) {
        const uint64_t s = *seed_state;
        s0 ^= (uint32_t)s * 0x9E3779B1u;
        s1 += (uint32_t)(s >> 32) * 0x85EBCA77u + (uint32_t)s;
    }
    return Drop{s0, s1, (uint32_t)(p_drop * 16777216.0f), 1.0f / (1.0f - p_drop)};
}

// acc[i] = b[row0 + rowof(i, h)]: the bias enters through the accumulator's initial value
__device__ __forceinline__ f32x16 bias_tile(const float *b, int row0, int h) {
    f32x16 a;
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(b + row0 + 8 * g + 4 * h);
        for (int q = 0; q < 4; ++q) a[4 * g + q] = v[q];
    }
    return a;
}
__device__ __forceinline__ f32x16 zero_tile() {
    f32x16 a;
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
    return a;
}
// acc += W[row0 .. row0+31][k0 .. k0 + 16 NK) . X, X = NK operand fragments (rows on lanes).  W row-major with leading
// dimension ld (elements); the A fragment of k-step ks is 16 bytes of row row0 + r at column k0 + 16 ks + 8 h.
template <int NK>
__device__ __forceinline__ f32x16 tile_gemm(const __bf16 *__restrict__ W, int ld, int row0, int k0, const bf16x8 *xf, f32x16 acc, int r,
                                            int h) {
    const __bf16 *p = W + (size_t)(row0 + r) * ld + k0 + 8 * h;
    bf16x8 a[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(p + 16 * ks);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) acc = mfma(a[ks], xf[ks], acc);
    return acc;
}
// operand fragments of an LDS activation tile (row-major bf16, byte stride `stride`): columns k0 .. k0 + 16 NK of row r
template <int NK>
__device__ __forceinline__ void load_frags(const char *buf, int stride, int k0, bf16x8 *xf, int r, int h) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) xf[ks] = *reinterpret_cast<const bf16x8 *>(buf + r * stride + 2 * (k0 + 16 * ks + 8 * h));
}
// four consecutive features (accumulator group g) of row r into a row-major LDS tile
__device__ __forceinline__ void put4(char *buf, int stride, int r, int col, const float v[4]) {
    bf16x4 pk;
    for (int q = 0; q < 4; ++q) pk[q] = (__bf16)v[q];
    *reinterpret_cast<bf16x4 *>(buf + r * stride + 2 * col) = pk;
}
// Fragment-packed layout of a bf16 matrix X[rows][cols] (rows % 32 == 0, cols % 16 == 0), the order in which a wavefront reads
// it as an MFMA operand: for every 32-row tile and every 16-column k-step, 64 lanes x 16 bytes = 1 KB contiguous,
//   offset(row, col) = ((((row / 32) * (cols / 16) + col / 16) * 2 + (col / 8) % 2) * 32 + row % 32) * 8 + col % 8.
// Row-major operands make every lane of a fragment load touch a different cache line (lane = row): 64 requests of 16 bytes per
// instruction, measured ~8 B/clk per CU; packed, one instruction is one contiguous KB.
__device__ __forceinline__ int64_t packed_off(int row, int64_t col, int64_t cols) {
    return ((((int64_t)(row >> 5) * (cols >> 4) + (col >> 4)) * 2 + ((col >> 3) & 1)) * 32 + (row & 31)) * 8 + (col & 7);
}
constexpr int RING = 3, DIST = 2;  // RING = DIST + 1: the slot of unit i + DIST was last read by unit i - 1

template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// LDS-only barrier: __syncthreads() would also wait for the weight fetches that are meant to stay in flight (vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sched_fence();
}
struct Ring {
    bf16x8 a[RING][16];
};
// fragments 0..7 at p + 512 ks, fragments 8..15 at p + off2 + 512 ks (elements; p = this lane's 16 bytes of the first fragment of a
// fragment-packed weight: one contiguous KB per wave-instruction)
template <int SLOT>
__device__ __forceinline__ void fetch_unit(Ring &R, const __bf16 *p, int64_t off2) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) R.a[SLOT][ks] = *reinterpret_cast<const bf16x8 *>(p + 512 * ks);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) R.a[SLOT][8 + ks] = *reinterpret_cast<const bf16x8 *>(p + off2 + 512 * ks);
}
template <int SLOT>
__device__ __forceinline__ f32x16 mm16(const Ring &R, const bf16x8 *xf, f32x16 acc) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc = mfma(R.a[SLOT][ks], xf[ks], acc);
    return acc;
}
template <int SLOT, int HALF>
__device__ __forceinline__ f32x16 mm8(const Ring &R, const bf16x8 *xf, f32x16 acc) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = mfma(R.a[SLOT][8 * HALF + ks], xf[ks], acc);
    return acc;
}
// per-lane address of a unit's first fragment in a fragment-packed [rows][cols] weight: row tile row0 / 32, k-step k0 / 16
__device__ __forceinline__ const __bf16 *unit_ptr(const void *W, int cols, int row0, int k0, int lane) {
    return (const __bf16 *)W + ((size_t)(row0 >> 5) * (cols >> 4) + (k0 >> 4)) * 512 + lane * 8;
}
constexpr int64_t NEXT_8_STEPS = 8 * 512;  // off2 of a unit = one row tile over 256 columns
__device__ __forceinline__ int64_t next_row_tile(int cols) { return (int64_t)(cols >> 4) * 512; }  // off2 of a unit = two row tiles over 128 columns

}  // namespace g2048_mfma
#endif  // G2048_MFMA_H
