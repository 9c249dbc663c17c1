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

template <typename T> struct RawV;                         // 4 elements of T kept packed until they are consumed
template <> struct RawV<float> {
    f32x4 v;
    FW_MEM void load(const float* p) { v = *reinterpret_cast<const f32x4*>(p); }
    FW_MEM void unpack(float* f) const { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
};
template <> struct RawV<bf16raw> {
    uint2 v;
    FW_MEM void load(const bf16raw* p) { v = *reinterpret_cast<const uint2*>(p); }
    FW_MEM void unpack(float* f) const {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};

// Both kernels are pure streams (read a row, two group reductions, write a row), i.e. bound by how many bytes a CU keeps in
// flight: with ONE row per lane group outstanding (the first form of these kernels) a CU had ~40 KB requested at a time and the
// pass ran at 2.8 TB/s.  Now a lane group owns U rows per step -- all their loads are issued back to back from CLAMPED coordinates
// (no branch between them), then the rows are reduced and written -- and the per-lane arrays are sized by NV = ceil(C / 4 / G),
// not by the C = 1024 maximum.  G lanes per row (16 / 32 / 64 by width) so that narrow rows (C = 28, 56) do not idle a wave.
template <typename T, int G, int NV, int U>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, long ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int C, float eps) {
    constexpr int RPB = 256 / G;
    const int gl = threadIdx.x % G, gr = threadIdx.x / G;
    const int nv = C >> 2;
    const long r0 = (long)blockIdx.x * RPB * U;
    f32x4 v[U][NV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long row = r0 + u * RPB + gr;
        const float* xr = x + (row < rows ? row : rows - 1) * ldx;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int c4 = gl + G * t;
            v[u][t] = *reinterpret_cast<const f32x4*>(xr + (c4 < nv ? c4 : nv - 1) * 4);
        }
    }
    f32x4 gam[NV], bet[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c4 = gl + G * t < nv ? gl + G * t : nv - 1;
        gam[t] = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
        bet[t] = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
    }
    const float invc = 1.0f / C;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long row = r0 + u * RPB + gr;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (gl + G * t >= nv) v[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[u][t][0] + v[u][t][1] + v[u][t][2] + v[u][t][3];
        }
        const float mu = group_sum<G>(s) * invc;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (gl + G * t < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[u][t][e] - mu; q += d * d; }
            }
        }
        const float rs = rsqrtf(group_sum<G>(q) * invc + eps);
        if (row >= rows) continue;
        if (gl == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int c4 = gl + G * t;
            if (c4 < nv) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[u][t][e] - mu) * rs * gam[t][e] + bet[t][e];
                T* yp = y + row * ldy + c4 * 4;
                if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(yp) = f32x4{o[0], o[1], o[2], o[3]};
                else *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// Each block walks row groups blockIdx.x, blockIdx.x + gridDim.x, ... of RPB * U rows; column partial sums stay in registers.
template <typename T, int G, int NV, int U>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres, long lddres,
                                                     float* __restrict__ dx, long lddx, float* __restrict__ partial,
                                                     long pstride, int rows, int C, T* __restrict__ twin, long ldtw,
                                                     const float* __restrict__ twscale, int tw_rows_per_scale) {
    constexpr int RPB = 256 / G;
    const int gl = threadIdx.x % G, gr = threadIdx.x / G;
    const int nv = C >> 2;
    const float invc = 1.0f / C;
    f32x4 gam[NV], dg[NV], db[NV];
    bool cok[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const int c4 = gl + G * t;
        cok[t] = c4 < nv;
        gam[t] = cok[t] ? *reinterpret_cast<const f32x4*>(gamma + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long r0 = (long)blockIdx.x * RPB * U; r0 < rows; r0 += (long)gridDim.x * RPB * U) {
        // ---- every load of the U rows first ----
        f32x4 xv[U][NV], rr[U][NV];
        RawV<T> dr[U][NV];
        float mu[U], rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = r0 + u * RPB + gr;
            const long rc = row < rows ? row : rows - 1;
            mu[u] = mean[rc]; rs[u] = rstd[rc];
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int cc = (cok[t] ? gl + G * t : nv - 1) * 4;
                xv[u][t] = *reinterpret_cast<const f32x4*>(x + rc * ldx + cc);
                dr[u][t].load(dy + rc * lddy + cc);
                if (dres) rr[u][t] = *reinterpret_cast<const f32x4*>(dres + rc * lddres + cc);
            }
        }
        // ---- then row by row ----
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = r0 + u * RPB + gr;
            const bool live = row < rows;
            f32x4 xh[NV], g[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                float d[4];
                dr[u][t].unpack(d);
                const bool ok = live && cok[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float de = ok ? d[e] : 0.f;
                    xh[t][e] = ok ? (xv[u][t][e] - mu[u]) * rs[u] : 0.f;
                    g[t][e] = de * gam[t][e];
                    dg[t][e] += de * xh[t][e];
                    db[t][e] += de;
                    s1 += g[t][e];
                    s2 += g[t][e] * xh[t][e];
                }
            }
            s1 = group_sum<G>(s1) * invc;
            s2 = group_sum<G>(s2) * invc;
            if (!live) continue;
            const float sc = (twin && twscale) ? twscale[row / tw_rows_per_scale] : 1.0f;
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                if (!cok[t]) continue;
                const int c4 = gl + G * t;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs[u] * (g[t][e] - s1 - xh[t][e] * s2);
                if (dres) o += rr[u][t];
                *reinterpret_cast<f32x4*>(dx + row * lddx + c4 * 4) = o;
                if (twin) {                                  // the operand copy the preceding Linear's backward needs: T(dx * DropPath scale)
                    T* tp = twin + row * ldtw + c4 * 4;
                    if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(tp) = o * sc;
                    else *reinterpret_cast<uint2*>(tp) = make_uint2(pack_bf2(o[0] * sc, o[1] * sc), pack_bf2(o[2] * sc, o[3] * sc));
                }
            }
        }
    }
    // block reduction of the column partials over the RPB row-groups, then one plain store per column
    __shared__ float red[2][1024];
    for (int i = threadIdx.x; i < 2048; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NV; ++t) {
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

// rows per lane group and step (U) by the number of 16-byte vectors a lane holds per row (NV): about 4 row-vectors in flight
template <typename T, int G, int NV, int U>
int ln_fwd_launch(const float* x, long ldx, const float* g, const float* b, void* y, long ldy, float* mean,
                  float* rstd, int rows, int C, float eps, hipStream_t st) {
    const int rpb = 256 / G * U;
    hipLaunchKernelGGL((ln_fwd_kernel<T, G, NV, U>), dim3(fw_cdiv(rows, rpb)), dim3(256), 0, st, x, ldx, g, b, (T*)y, ldy,
                       mean, rstd, rows, C, eps);
    FW_LAUNCH_RET();
}
static inline int ln_group(int C) { return C <= 32 ? 8 : (C <= 64 ? 16 : (C <= 128 ? 32 : 64)); }
static inline int ln_rows_per_step(int C) { return C <= 256 ? 4 : (C <= 512 ? 2 : 1); }
// Blocks of the backward pass: every block ends in a fold of its column partials (2 C LDS atomics per lane group, 2 C floats written
// for fw_slab_reduce to re-read), so wide rows want FEW blocks that walk many rows -- rocprofv3 averages at caps 1024 / 512 / 256:
// C = 896: 36.2 / 26.7 / 22.1 us, C = 448: 25.6 / 23.4 / 22.3 us, C = 224: 25.7 / 24.4 / 27.9 us, C <= 112: 46.5 / 47.4 / 57.8 us.
static inline int ln_bwd_grid(int rows, int C) {
    static const int env_cap = getenv("FW_LN_BWD_GRID") ? atoi(getenv("FW_LN_BWD_GRID")) : 0;
    const int cap = env_cap > 0 ? env_cap : (C > 256 ? 256 : (C > 128 ? 512 : 1024));
    return min(fw_cdiv(rows, 256 / ln_group(C) * ln_rows_per_step(C)), cap);
}

template <typename T, int G, int NV, int U>
int ln_bwd_launch(const void* dy, long lddy, const float* x, long ldx, const float* g, const float* mean,
                  const float* rstd, const float* dres, long lddres, float* dx, long lddx, float* partial, long pstride,
                  int rows, int C, void* twin, long ldtw, const float* twscale, int twrps, hipStream_t st) {
    hipLaunchKernelGGL((ln_bwd_kernel<T, G, NV, U>), dim3(ln_bwd_grid(rows, C)), dim3(256), 0, st, (const T*)dy, lddy, x, ldx, g, mean, rstd,
                       dres, lddres, dx, lddx, partial, pstride, rows, C, (T*)twin, ldtw, twscale, twrps);
    FW_LAUNCH_RET();
}

// (G, NV, U) by row width
#define LN_DISPATCH(FN, T, ...)                                    \
    (C <= 32 ? FN<T, 8, 1, 4>(__VA_ARGS__)                         \
     : C <= 64 ? FN<T, 16, 1, 4>(__VA_ARGS__)                      \
     : C <= 128 ? FN<T, 32, 1, 4>(__VA_ARGS__)                     \
     : C <= 256 ? FN<T, 64, 1, 4>(__VA_ARGS__)                     \
     : C <= 512 ? FN<T, 64, 2, 2>(__VA_ARGS__)                     \
                : FN<T, 64, 4, 1>(__VA_ARGS__))

}  // namespace

extern "C" int fw_layernorm_fwd(int dtype, const float* x, long ldx, const float* gamma, const float* beta, void* y,
                                long ldy, float* mean, float* rstd, int rows, int C, float eps, void* stream) {
    FW_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0);
    FW_CHECK_ARG(x && gamma && beta && y && mean && rstd);
    hipStream_t st = (hipStream_t)stream;
#define LN_F(T) LN_DISPATCH(ln_fwd_launch, T, x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C, eps, st)
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
#define LN_B(T)                                                                                                          \
    LN_DISPATCH(ln_bwd_launch, T, dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dx, lddx, partial, pstride, rows, C, twin, \
                ldtw, twscale, tw_rows_per_scale, st)
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
