// LayerNorm forward / backward over the fp32 residual stream (K3 prologue of SURVEY.md 2.2).
// Replaces nn.LayerNorm at decoder_Uformer.py:567,594,666,744 / encoder_Uformer.py:549,577,641,680
// and the LN in front of the encoder heads (encoder_Uformer.py:941).  HBM-bound: one pass, 16-byte
// loads, G lanes per row (G = 16/32/64 by width) so narrow rows (C = 28, 56) do not idle a wave.
//   fwd : y[T] = (x - mean) * rstd * gamma + beta ; saves mean, rstd (fp32)
//   bwd : dx[f32] = (dres?) + rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * gamma
//         dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy      (block partials -> atomicAdd)
#include "fw_common.h"
#include <stdlib.h>

namespace {

constexpr int NVMAX = 4;   // float4 per lane: C <= 4*4*64 = 1024

template <int G> FW_DEV float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int G>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int C, float eps) {
    const int gl = threadIdx.x % G;
    const int row = (int)((blockIdx.x * 256L + threadIdx.x) / G);
    const bool live = row < rows;
    const int nv = C >> 2;
    f32x4 v[NVMAX];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NVMAX; ++t) {
        const int c4 = gl + G * t;
        v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live && c4 < nv) v[t] = *reinterpret_cast<const f32x4*>(x + (long)row * ldx + c4 * 4);
        s += v[t][0] + v[t][1] + v[t][2] + v[t][3];
    }
    const float mu = group_sum<G>(s) / C;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NVMAX; ++t) {
        const int c4 = gl + G * t;
        if (c4 < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[t][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(group_sum<G>(q) / C + eps);
    if (!live) return;
    if (gl == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int t = 0; t < NVMAX; ++t) {
        const int c4 = gl + G * t;
        if (c4 < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[t][e] - mu) * rs * g[e] + b[e];
            T* yp = y + (long)row * ldy + c4 * 4;
            if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
            else *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
        }
    }
}

template <typename T> FW_DEV void load4(const T* p, float* f);
template <> FW_SPEC void load4<float>(const float* p, float* f) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> FW_SPEC void load4<bf16raw>(const bf16raw* p, float* f) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

// Each block walks rows blockIdx.x*RPB + k*gridDim.x*RPB ...; column partial sums stay in registers.
template <typename T, int G>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres, long lddres,
                                                     float* __restrict__ dx, long lddx, float* __restrict__ partial,
                                                     long pstride, int rows, int C, T* __restrict__ twin, long ldtw,
                                                     const float* __restrict__ twscale, int tw_rows_per_scale) {
    constexpr int RPB = 256 / G;
    const int gl = threadIdx.x % G, gr = threadIdx.x / G;
    const int nv = C >> 2;
    f32x4 gam[NVMAX], dg[NVMAX], db[NVMAX];
#pragma unroll
    for (int t = 0; t < NVMAX; ++t) {
        const int c4 = gl + G * t;
        gam[t] = (c4 < nv) ? *reinterpret_cast<const f32x4*>(gamma + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long r0 = (long)blockIdx.x * RPB; r0 < rows; r0 += (long)gridDim.x * RPB) {
        const long row = r0 + gr;
        const bool live = row < rows;
        const float mu = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
        f32x4 xh[NVMAX], g[NVMAX];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NVMAX; ++t) {
            const int c4 = gl + G * t;
            xh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            g[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && c4 < nv) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * ldx + c4 * 4);
                float d[4];
                load4<T>(dy + row * lddy + c4 * 4, d);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[t][e] = (xv[e] - mu) * rs;
                    g[t][e] = d[e] * gam[t][e];
                    dg[t][e] += d[e] * xh[t][e];
                    db[t][e] += d[e];
                    s1 += g[t][e];
                    s2 += g[t][e] * xh[t][e];
                }
            }
        }
        s1 = group_sum<G>(s1) / C;
        s2 = group_sum<G>(s2) / C;
        if (live) {
#pragma unroll
            for (int t = 0; t < NVMAX; ++t) {
                const int c4 = gl + G * t;
                if (c4 < nv) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rs * (g[t][e] - s1 - xh[t][e] * s2);
                    if (dres) {
                        const f32x4 rr = *reinterpret_cast<const f32x4*>(dres + row * lddres + c4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += rr[e];
                    }
                    *reinterpret_cast<f32x4*>(dx + row * lddx + c4 * 4) = o;
                    if (twin) {                                  // the operand copy the preceding Linear's backward needs: T(dx * DropPath scale)
                        const float sc = twscale ? twscale[row / tw_rows_per_scale] : 1.0f;
                        T* tp = twin + row * ldtw + c4 * 4;
                        if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(tp) = f32x4{o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc};
                        else *reinterpret_cast<uint2*>(tp) = make_uint2(pack_bf2(o[0] * sc, o[1] * sc), pack_bf2(o[2] * sc, o[3] * sc));
                    }
                }
            }
        }
    }
    // block reduction of the column partials over the RPB row-groups, then one atomic per column
    __shared__ float red[2][1024];
    for (int i = threadIdx.x; i < 2048; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NVMAX; ++t) {
        const int c4 = gl + G * t;
        if (c4 < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(&red[0][c4 * 4 + e], dg[t][e]);
                atomicAdd(&red[1][c4 * 4 + e], db[t][e]);
            }
        }
    }
    __syncthreads();
    // block partials [block][dgamma(Cp) | dbeta(Cp)]: plain stores; fw_slab_reduce folds them (thousands of blocks adding
    // atomically into the same C words serialise in L2 and cost more than the whole LayerNorm pass)
    float* pp = partial + (long)blockIdx.x * pstride;
    const int Cp = (int)(pstride >> 1);
    for (int c = threadIdx.x; c < Cp; c += 256) {
        pp[c] = c < C ? red[0][c] : 0.f;
        pp[Cp + c] = c < C ? red[1][c] : 0.f;
    }
}

template <typename T, int G>
int ln_fwd_launch(const float* x, long ldx, const float* g, const float* b, void* y, long ldy, float* mean,
                  float* rstd, int rows, int C, float eps, hipStream_t st) {
    const int rpb = 256 / G;
    hipLaunchKernelGGL((ln_fwd_kernel<T, G>), dim3(fw_cdiv(rows, rpb)), dim3(256), 0, st, x, ldx, g, b, (T*)y, ldy,
                       mean, rstd, rows, C, eps);
    FW_LAUNCH_RET();
}
static inline int ln_group(int C) { return C <= 64 ? 16 : (C <= 128 ? 32 : 64); }
static inline int ln_bwd_grid(int rows, int C) {
    static const int cap = getenv("FW_LN_BWD_GRID") ? atoi(getenv("FW_LN_BWD_GRID")) : 1024;
    return min(fw_cdiv(rows, 256 / ln_group(C)), cap);
}

template <typename T, int G>
int ln_bwd_launch(const void* dy, long lddy, const float* x, long ldx, const float* g, const float* mean,
                  const float* rstd, const float* dres, long lddres, float* dx, long lddx, float* partial, long pstride,
                  int rows, int C, void* twin, long ldtw, const float* twscale, int twrps, hipStream_t st) {
    hipLaunchKernelGGL((ln_bwd_kernel<T, G>), dim3(ln_bwd_grid(rows, C)), dim3(256), 0, st, (const T*)dy, lddy, x, ldx, g, mean, rstd,
                       dres, lddres, dx, lddx, partial, pstride, rows, C, (T*)twin, ldtw, twscale, twrps);
    FW_LAUNCH_RET();
}

}  // namespace

extern "C" int fw_layernorm_fwd(int dtype, const float* x, long ldx, const float* gamma, const float* beta, void* y,
                                long ldy, float* mean, float* rstd, int rows, int C, float eps, void* stream) {
    FW_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0);
    FW_CHECK_ARG(x && gamma && beta && y && mean && rstd);
    hipStream_t st = (hipStream_t)stream;
#define LN_F(T)                                                                                                 \
    (C <= 64 ? ln_fwd_launch<T, 16>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C, eps, st)                  \
             : C <= 128 ? ln_fwd_launch<T, 32>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C, eps, st)       \
                        : ln_fwd_launch<T, 64>(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C, eps, st))
    return dtype == FW_DT_BF16 ? LN_F(bf16raw) : LN_F(float);
#undef LN_F
}

// Number of block partials fw_layernorm_bwd writes for (rows, C): the caller provides `partial` f32 [blocks][2 * roundup(C, 4)].
extern "C" int fw_layernorm_bwd_blocks(int rows, int C) { return ln_bwd_grid(rows, C); }

// dx = (dres?) + LN'(dy).  The per-block column sums of dy*xhat / dy go to `partial` (plain stores, row = [dgamma | dbeta]) and
// are folded into dgamma / dbeta (accumulated) by fw_slab_reduce, launched here on the same stream -- unless both are null.
extern "C" int fw_slab_reduce(const float* slab, int nz, long n, long zstride, float* dst, int accumulate, float* dst2, long off2, long n2,
                              void* stream);
extern "C" int fw_layernorm_bwd2(int dtype, const void* dy, long lddy, const float* x, long ldx, const float* gamma,
                                 const float* mean, const float* rstd, const float* dres, long lddres, float* dx,
                                 long lddx, float* dgamma, float* dbeta, float* partial, int rows, int C, void* twin, long ldtw,
                                 const float* twscale, int tw_rows_per_scale, void* stream) {
    FW_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0);
    FW_CHECK_ARG(dy && x && gamma && mean && rstd && dx && partial && ((dgamma && dbeta) || (!dgamma && !dbeta)));
    FW_CHECK_ARG(!twin || (ldtw % 4 == 0 && (!twscale || tw_rows_per_scale > 0)));
    hipStream_t st = (hipStream_t)stream;
    const long pstride = 2L * C;
#define LN_B(T)                                                                                                      \
    (C <= 64 ? ln_bwd_launch<T, 16>(dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, partial, pstride, rows, \
                                    C, twin, ldtw, twscale, tw_rows_per_scale, st)                                                                           \
             : C <= 128 ? ln_bwd_launch<T, 32>(dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, partial,  \
                                               pstride, rows, C, twin, ldtw, twscale, tw_rows_per_scale, st)                                                 \
                        : ln_bwd_launch<T, 64>(dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, partial,  \
                                               pstride, rows, C, twin, ldtw, twscale, tw_rows_per_scale, st))
    const int rc = dtype == FW_DT_BF16 ? LN_B(bf16raw) : LN_B(float);
#undef LN_B
    if (rc || !dgamma) return rc;                  // dgamma == dbeta == null: the caller folds the partials later (fw_slab_reduce_multi)
    return fw_slab_reduce(partial, ln_bwd_grid(rows, C), C, pstride, dgamma, 1, dbeta, C, C, stream);
}

extern "C" int fw_layernorm_bwd(int dtype, const void* dy, long lddy, const float* x, long ldx, const float* gamma,
                                const float* mean, const float* rstd, const float* dres, long lddres, float* dx,
                                long lddx, float* dgamma, float* dbeta, float* partial, int rows, int C, void* stream) {
    return fw_layernorm_bwd2(dtype, dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, dgamma, dbeta, partial, rows, C,
                             nullptr, 0, nullptr, 1, stream);
}
