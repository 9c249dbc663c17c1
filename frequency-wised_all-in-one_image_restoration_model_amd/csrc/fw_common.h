// Shared device primitives for the gfx950 (MI355X / CDNA4) kernels of the AirNet hot path.
//
// Everything here is written for 64-lane wavefronts and the CDNA4 MFMA fragment maps:
//   v_mfma_f32_16x16x32_bf16 : A lane l holds A[row l&15][k = 8*(l>>4) + j], j = 0..7 (16 B)
//                              B lane l holds B[k = 8*(l>>4) + j][col l&15]
//   v_mfma_f32_16x16x4_f32   : A lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15]
//   C/D (both)               : col = l&15, row = 4*(l>>4) + reg
// Operand tiles live in LDS as row-major [rows][K] with K contiguous ("k-contiguous").  In BYTES
// the bf16 and f32 paths are identical: one "k-chunk" is 64 bytes of K per row, a lane's fragment
// is the 16 bytes at offset 16*(l>>4) inside the chunk.  bf16 consumes a chunk with ONE
// 16x16x32 MFMA; f32 consumes it with FOUR 16x16x4 MFMAs (step s uses dword s of the 16 B, i.e.
// k = 4*(l>>4)+s -- the same slot permutation on both operands, so the sum is unchanged).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#define FW_DT_F32 0
#define FW_DT_BF16 1

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef unsigned short bf16raw;

#define FW_DEV static __device__ __forceinline__
#define FW_SPEC __device__ __forceinline__      /* explicit specialisations */
#define FW_MEM __device__ __forceinline__       /* non-static member functions */

FW_DEV int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------- scalar conversions
FW_DEV float bf2f(bf16raw b) { return __uint_as_float(((unsigned)b) << 16); }
FW_DEV bf16raw f2bf(float f) {
    __bf16 h = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16raw, h);
}
// two floats -> one dword of bf16: ONE v_cvt_pk_bf16_f32 (RNE, NaN stays NaN); converting the halves separately costs
// two conversions and an OR per pair, and the pair conversion runs in every epilogue and every LDS transpose store
typedef __attribute__((ext_vector_type(2))) float fw_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 fw_bf16x2;
FW_DEV unsigned pack_bf2(float lo, float hi) {
    const fw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, fw_bf16x2));
}

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int SZ = 4;
    static constexpr int E16 = 4;               // elements per 16 bytes
    FW_DEV float ld(const float* p) { return *p; }
    FW_DEV void st(float* p, float v) { *p = v; }
};
template <> struct TT<bf16raw> {
    static constexpr int SZ = 2;
    static constexpr int E16 = 8;
    FW_DEV float ld(const bf16raw* p) { return bf2f(*p); }
    FW_DEV void st(bf16raw* p, float v) { *p = f2bf(v); }
};

// unpack / pack 16 bytes <-> floats
template <typename T> FW_DEV void unpack16(const uint4& v, float* f);
template <> FW_SPEC void unpack16<float>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
    f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> FW_SPEC void unpack16<bf16raw>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> FW_DEV uint4 pack16(const float* f);
template <> FW_SPEC uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> FW_SPEC uint4 pack16<bf16raw>(const float* f) {
    return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// ---------------------------------------------------------------- math
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. f32 rounding level): ~12 VALU + one v_exp, branch-free.
// libm's erff is a multi-branch polynomial several times that long, and GELU / GELU' run on every hidden element.
FW_DEV float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp), not the 10-instruction IEEE division
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
FW_DEV float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
FW_DEV float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// The bf16 instantiations evaluate GELU / GELU' WITHOUT transcendentals: x is clamped to [-4, 4] and Phi(x) - 1/2 = x P(x^2),
// GELU'(x) - 1/2 = x R(x^2) with degree-7 P, R (tools/fit_gelu_poly.py: weighted minimax fits; |GELU error| <= 5e-5 on [-4, 4] and
// <= 4e-5 |x| beyond, |GELU' error| <= 3e-4 -- the results are then rounded to bf16, whose half-ulp is 2e-3 |x|).  11 full-rate
// VALU instructions that pair up as v_pk_fma_f32, against ~14 plus two quarter-rate transcendentals (v_rcp, v_exp) of the erf form:
// the fc1 / fc2-input-gradient epilogues and the depthwise kernels were bound by exactly this arithmetic.  f32 keeps the erf form.
FW_DEV float gelu_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float u = xc * xc;
    float p = -1.301290697e-09f;
    p = fmaf(p, u, 1.041962340e-07f); p = fmaf(p, u, -3.657149647e-06f); p = fmaf(p, u, 7.485542814e-05f);
    p = fmaf(p, u, -1.006494109e-03f); p = fmaf(p, u, 9.505412949e-03f); p = fmaf(p, u, -6.588785358e-02f);
    p = fmaf(p, u, 3.986733717e-01f);
    return x * fmaf(xc, p, 0.5f);
}
FW_DEV float gelu_grad_poly(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float u = xc * xc;
    float r = -1.641741003e-08f;
    r = fmaf(r, u, 1.213663647e-06f); r = fmaf(r, u, -3.845618281e-05f); r = fmaf(r, u, 6.876052189e-04f);
    r = fmaf(r, u, -7.687198903e-03f); r = fmaf(r, u, 5.591419208e-02f); r = fmaf(r, u, -2.620297180e-01f);
    r = fmaf(r, u, 7.967230048e-01f);
    return fmaf(xc, r, 0.5f);
}
template <typename T> FW_DEV float gelu_t(float x) { return sizeof(T) == 2 ? gelu_poly(x) : gelu_f(x); }
template <typename T> FW_DEV float gelu_grad_t(float x) { return sizeof(T) == 2 ? gelu_grad_poly(x) : gelu_grad_f(x); }
FW_DEV float lrelu_f(float x, float s) { return x > 0.f ? x : x * s; }

FW_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
FW_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- counter-based Bernoulli masks (nn.Dropout, encoder_ViT.py:31-33,67,73,158)
// A mask element is a pure function of (seed, site, element index): nothing is stored between forward and backward, the
// backward pass re-derives the forward's mask, and the CPU oracle (oracle/dropout_hash.py) evaluates the same integers.
// seed: one u32 in device memory, advanced once per training step (fw_rng_tick); site: one id per Dropout module call site.
FW_DEV unsigned fw_hash32(unsigned x) {                       // "lowbias32" integer finaliser
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
FW_DEV unsigned fw_site_key(unsigned seed, unsigned site) { return fw_hash32(seed ^ fw_hash32(site * 0x9E3779B9U + 0x7F4A7C15U)); }
// keep with probability 1 - thresh / 2^32
FW_DEV bool fw_keep(unsigned key, unsigned long long idx, unsigned thresh) {
    return fw_hash32(fw_hash32((unsigned)idx ^ key) + (unsigned)(idx >> 32)) >= thresh;
}
static inline unsigned fw_drop_thresh(float p) {
    const double t = (double)p * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
}

// ---------------------------------------------------------------- MFMA on one 64-byte k-chunk
template <typename T> FW_DEV void mma_chunk(f32x4& acc, const uint4& a, const uint4& b);
template <> FW_SPEC void mma_chunk<bf16raw>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                  __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> FW_SPEC void mma_chunk<float>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Fragment of a k-contiguous LDS tile: rows row0..row0+15, chunk c (64 B of K), row stride ldb BYTES.
FW_DEV uint4 frag_kc(const char* tile, int ldb, int row0, int chunk) {
    const int l = lane_id();
    return *reinterpret_cast<const uint4*>(tile + (row0 + (l & 15)) * ldb + chunk * 64 + ((l >> 4) << 4));
}

// Fragment of a K-MAJOR LDS tile (stored [k][m], m contiguous, row stride ldb bytes): the operand
// rows m0..m0+15 over k-chunk c (k = c*KE .. c*KE+KE-1, KE = 32 for bf16 / 16 for f32).
template <typename T> FW_DEV uint4 frag_km(const char* tile, int ldb, int m0, int chunk);
template <> FW_SPEC uint4 frag_km<float>(const char* tile, int ldb, int m0, int chunk) {
    const int l = lane_id();
    const char* p = tile + (chunk * 16 + ((l >> 4) << 2)) * ldb + (m0 + (l & 15)) * 4;
    uint4 r;
    r.x = *reinterpret_cast<const unsigned*>(p);
    r.y = *reinterpret_cast<const unsigned*>(p + ldb);
    r.z = *reinterpret_cast<const unsigned*>(p + 2 * ldb);
    r.w = *reinterpret_cast<const unsigned*>(p + 3 * ldb);
    return r;
}
// bf16: gfx950's transposing LDS read.  ds_read_b64_tr_b16 hands lane i of each 16-lane group column i of a 4-row x
// 16-column block (row q in element q); lane 4q+p supplies the address of row q, columns 4p..4p+3.  Two reads (k rows
// 8g..8g+3 and 8g+4..8g+7 of the chunk) are exactly the 8 k-values of the 16x16x32 fragment.  Needs EXEC = all ones
// (call from wave-uniform code only), 8-byte aligned addresses (ldb % 8 == 0, m0 % 4 == 0).
typedef __attribute__((ext_vector_type(4))) short fw_s16x4;
typedef __attribute__((address_space(3))) fw_s16x4 fw_lds_s16x4;
template <> FW_SPEC uint4 frag_km<bf16raw>(const char* tile, int ldb, int m0, int chunk) {
    const int l = lane_id();
    const char* p = tile + (chunk * 32 + ((l >> 4) << 3) + ((l >> 2) & 3)) * ldb + (m0 + ((l & 3) << 2)) * 2;
    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((fw_lds_s16x4*)p));
    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((fw_lds_s16x4*)(p + 4 * ldb)));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// Store one C/D tile.  acc element r of lane l is C[row0 + 4*(l>>4) + r][col0 + (l&15)].
// store_acc_T writes the TRANSPOSE, dst[col][row] (4 consecutive rows -> one 8/16-byte write);
// store_acc_N writes dst[row][col] element by element.  ldb = row stride of dst in BYTES.
template <typename T> FW_DEV void store_acc_T(char* dst, int ldb, int row0, int col0, const f32x4& acc);
template <> FW_SPEC void store_acc_T<float>(char* dst, int ldb, int row0, int col0, const f32x4& acc) {
    const int l = lane_id();
    *reinterpret_cast<f32x4*>(dst + (col0 + (l & 15)) * ldb + (row0 + ((l >> 4) << 2)) * 4) = acc;
}
template <> FW_SPEC void store_acc_T<bf16raw>(char* dst, int ldb, int row0, int col0, const f32x4& acc) {
    const int l = lane_id();
    uint2 v = make_uint2(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]));
    *reinterpret_cast<uint2*>(dst + (col0 + (l & 15)) * ldb + (row0 + ((l >> 4) << 2)) * 2) = v;
}
template <typename T> FW_DEV void store_acc_N(char* dst, int ldb, int row0, int col0, const f32x4& acc) {
    const int l = lane_id();
    char* p = dst + (row0 + ((l >> 4) << 2)) * ldb + (col0 + (l & 15)) * TT<T>::SZ;
#pragma unroll
    for (int r = 0; r < 4; ++r) TT<T>::st(reinterpret_cast<T*>(p + r * ldb), acc[r]);
}

// acc[MT][NT] += A(rows a_row0.., k-contiguous) * B(rows b_row0.., k-contiguous)^T over KC chunks.
template <typename T, int MT, int NT>
FW_DEV void mma_tiles(f32x4 (&acc)[MT][NT], const char* A, int lda, int a_row0, const char* B, int ldb,
                      int b_row0, int KC) {
    for (int c = 0; c < KC; ++c) {
        uint4 af[MT], bfr[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = frag_kc(A, lda, a_row0 + 16 * m, c);
#pragma unroll
        for (int n = 0; n < NT; ++n) bfr[n] = frag_kc(B, ldb, b_row0 + 16 * n, c);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
    }
}

template <int MT, int NT> FW_DEV void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---------------------------------------------------------------- host-side helpers
#define FW_CHECK_ARG(cond)                                                                   \
    do {                                                                                     \
        if (!(cond)) return -__LINE__;                                                       \
    } while (0)
#define FW_LAUNCH_RET()                                                                      \
    do {                                                                                     \
        hipError_t e__ = hipGetLastError();                                                  \
        return e__ == hipSuccess ? 0 : (int)e__;                                             \
    } while (0)

// Raise a kernel's dynamic-LDS limit exactly once per template instantiation.  Entry points are called from the Python main
// thread AND from autograd worker threads: an unsynchronised `static bool` raced there.
#define FW_SET_LDS_ONCE(kern, bytes)                                                         \
    do {                                                                                     \
        static std::once_flag once__;                                                        \
        std::call_once(once__, [&] {                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        });                                                                                  \
    } while (0)

static inline int fw_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
