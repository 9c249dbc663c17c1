// Small kernels of the ViT encoder plug-in (net/encoder_ViT.py:17-203, BASELINE configs[4]) that the shared GEMM / LayerNorm /
// attention kernels do not cover:
//   * pos_embedding add (encoder_ViT.py:187): out[r] = x[r] + pos[r % N]
//   * BatchNorm2d + LeakyReLU(0.1) + AdaptiveAvgPool2d(1) on the [B][ED][P] planes of `inter` (encoder_ViT.py:170-173,196-199) with
//     BOTH outputs materialised -- the normalised map `inter` is part of the encoder's return value -- and a backward that takes
//     gradients of both (the Uformer encoder's fused head kernel only ever needs the pooled vector)
//   * the encoder_dim x encoder_dim MLP of the contrastive head (encoder_ViT.py:175-179): encoder_dim defaults to 3 for the ViT
//     (option.py:80-101), far below one MFMA tile and not a multiple of the GEMM kernels' 4-column granule.
// All HBM-bound elementwise / reduction work on a few hundred KB: plain grid-stride kernels, f32 statistics.
#include "fw_common.h"

namespace {
constexpr int TPB = 256;
FW_DEV long gtid() { return (long)blockIdx.x * blockDim.x + threadIdx.x; }
FW_DEV long gstride() { return (long)gridDim.x * blockDim.x; }
static inline int grid_for(long n) { long g = (n + TPB - 1) / TPB; return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }

__global__ void add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ p, float* __restrict__ out, long n, long period) {
    for (long i = gtid(); i < n; i += gstride()) out[i] = x[i] + p[i % period];
}

FW_DEV float block_sum(float v, float* red) {               // 1024-thread blocks
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// one block per channel c: statistics over (b, p), then inter = lrelu((x - mean) rstd gamma + beta), gap[b][c] = mean_p inter
template <typename T>
__global__ __launch_bounds__(1024) void bn_planes_fwd_kernel(const T* __restrict__ fea, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* rmean, float* rvar, long long* nbt, float* __restrict__ mr,
                                                             float* __restrict__ inter, float* __restrict__ gap, int B, int ED, int P,
                                                             int training, float eps, float momentum, float slope) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const long M = (long)B * P;
    float mean, rstd;
    if (training) {
        const float pivot = TT<T>::ld(fea + (long)c * P);          // shifted sums: no cancellation when the spread is small against the mean
        float s = 0.f, q = 0.f;
        for (long i = threadIdx.x; i < M; i += blockDim.x) {
            const float v = TT<T>::ld(fea + ((i / P) * ED + c) * P + i % P) - pivot;
            s += v; q += v * v;
        }
        s = block_sum(s, red); q = block_sum(q, red);
        const float md = s / M;
        mean = pivot + md;
        const float var = fmaxf(q / M - md * md, 0.f);
        rstd = rsqrtf(var + eps);
        if (threadIdx.x == 0) {
            rmean[c] = rmean[c] * (1.f - momentum) + momentum * mean;
            rvar[c] = rvar[c] * (1.f - momentum) + momentum * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
            if (c == 0 && nbt) *nbt += 1;
        }
    } else {
        mean = rmean[c]; rstd = rsqrtf(rvar[c] + eps);
    }
    if (threadIdx.x == 0) { mr[c] = mean; mr[ED + c] = rstd; }
    const float g = gamma[c] * rstd, bb = beta[c] - mean * g;
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int p = threadIdx.x; p < P; p += blockDim.x) {
            const long o = ((long)b * ED + c) * P + p;
            const float v = lrelu_f(TT<T>::ld(fea + o) * g + bb, slope);
            inter[o] = v; s += v;
        }
        s = block_sum(s, red);
        if (threadIdx.x == 0) gap[(long)b * ED + c] = s / P;
    }
}
// dy = dinter + dgap / P;  dz = dy lrelu'(inter);  dfea = gamma rstd (dz - mean(dz) - xhat mean(dz xhat)) (train) | gamma rstd dz (eval)
template <typename T>
__global__ __launch_bounds__(1024) void bn_planes_bwd_kernel(const T* __restrict__ fea, const float* __restrict__ inter, const float* __restrict__ gamma,
                                                             const float* __restrict__ mr, const float* __restrict__ dinter,
                                                             const float* __restrict__ dgap, T* __restrict__ dfea, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int B, int ED, int P, int training, float slope) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const long M = (long)B * P;
    const float mean = mr[c], rstd = mr[ED + c];
    float s0 = 0.f, s1 = 0.f;
    for (long i = threadIdx.x; i < M; i += blockDim.x) {
        const long o = ((i / P) * ED + c) * P + i % P;
        float dz = (dinter ? dinter[o] : 0.f) + (dgap ? dgap[(i / P) * ED + c] / P : 0.f);
        if (inter[o] < 0.f) dz *= slope;
        s0 += dz; s1 += dz * (TT<T>::ld(fea + o) - mean) * rstd;
    }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red);
    if (threadIdx.x == 0) { dgamma[c] = s1; dbeta[c] = s0; }
    const float g = gamma[c] * rstd, inv = 1.0f / (float)M;
    for (long i = threadIdx.x; i < M; i += blockDim.x) {
        const long o = ((i / P) * ED + c) * P + i % P;
        float dz = (dinter ? dinter[o] : 0.f) + (dgap ? dgap[(i / P) * ED + c] / P : 0.f);
        if (inter[o] < 0.f) dz *= slope;
        const float xh = (TT<T>::ld(fea + o) - mean) * rstd;
        TT<T>::st(dfea + o, training ? g * (dz - s0 * inv - xh * s1 * inv) : g * dz);
    }
}

// y[m][n] = lrelu(sum_k x[m][k] w[n][k] + b[n], slope)   (slope 1 = identity); one thread per output element
__global__ void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                        int M, int N, int K, float slope) {
    for (long i = gtid(); i < (long)M * N; i += gstride()) {
        const int m = (int)(i / N), n = (int)(i % N);
        float s = b ? b[n] : 0.f;
        for (int k = 0; k < K; ++k) s += x[(long)m * K + k] * w[(long)n * K + k];
        y[i] = lrelu_f(s, slope);
    }
}
// dz = dy lrelu'(y);  dx[m][k] = sum_n dz w[n][k];  dw[n][k] = sum_m dz x[m][k];  db[n] = sum_m dz.   One launch, three index ranges.
__global__ void small_linear_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                        const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                        int M, int N, int K, float slope) {
    const long n1 = (long)M * K, n2 = (long)N * K, n3 = N;
    for (long i = gtid(); i < n1 + n2 + n3; i += gstride()) {
        float s = 0.f;
        if (i < n1) {
            const int m = (int)(i / K), k = (int)(i % K);
            for (int n = 0; n < N; ++n) { float dz = dy[(long)m * N + n]; if (y[(long)m * N + n] < 0.f) dz *= slope; s += dz * w[(long)n * K + k]; }
            dx[i] = s;
        } else if (i < n1 + n2) {
            const long j = i - n1; const int n = (int)(j / K), k = (int)(j % K);
            for (int m = 0; m < M; ++m) { float dz = dy[(long)m * N + n]; if (y[(long)m * N + n] < 0.f) dz *= slope; s += dz * x[(long)m * K + k]; }
            dw[j] = s;
        } else {
            const int n = (int)(i - n1 - n2);
            for (int m = 0; m < M; ++m) { float dz = dy[(long)m * N + n]; if (y[(long)m * N + n] < 0.f) dz *= slope; s += dz; }
            db[n] = s;
        }
    }
}

// ---- nn.Dropout call sites of the ViT (encoder_ViT.py:31,33,73,158,189) as counter-based masks (fw_common.h: fw_keep) -----------
// mode 0: y = drop(x)                          f32 -> f32   (backward of modes 1 and 4: the same mask on the gradient)
// mode 1: y = res + drop(x)                    f32          (to_out / FeedForward output + residual stream)
// mode 2: y = drop(gelu(x))                    T -> T       (FeedForward hidden: GELU, Dropout)
// mode 3: y = drop(x) * gelu'(aux)             T -> T       (backward of mode 2)
// mode 4: y = drop(x + aux[i % period])        f32          (x += pos_embedding; emb dropout)
// thresh == 0 (eval / p = 0) keeps everything.  The element index i is the flat index of the (contiguous) tensor.
template <typename T>
__global__ void dropout_kernel(int mode, const void* __restrict__ xv, const void* __restrict__ auxv, const float* __restrict__ res,
                               void* __restrict__ yv, long n, long period, const unsigned* __restrict__ seed, unsigned site, unsigned thresh,
                               float inv_keep) {
    const unsigned key = thresh ? fw_site_key(seed[0], site) : 0u;
    for (long i = gtid(); i < n; i += gstride()) {
        const float m = (!thresh || fw_keep(key, (unsigned long long)i, thresh)) ? inv_keep : 0.f;
        if (mode == 2 || mode == 3) {
            const T* x = (const T*)xv; T* y = (T*)yv;
            const float v = TT<T>::ld(x + i);
            TT<T>::st(y + i, mode == 2 ? gelu_t<T>(v) * m : v * m * gelu_grad_t<T>(TT<T>::ld((const T*)auxv + i)));
        } else {
            const float* x = (const float*)xv; float* y = (float*)yv;
            if (mode == 0) y[i] = x[i] * m;
            else if (mode == 1) y[i] = res[i] + x[i] * m;
            else y[i] = (x[i] + ((const float*)auxv)[i % period]) * m;
        }
    }
}
__global__ void rng_tick_kernel(unsigned* seed) { if (gtid() == 0) seed[0] += 1u; }
}  // namespace

#define ST ((hipStream_t)stream)
extern "C" int fw_dropout(int mode, int dtype, const void* x, const void* aux, const float* res, void* y, long n, long period, const void* seed,
                          int site, float p, void* stream) {
    FW_CHECK_ARG(x && y && n > 0 && mode >= 0 && mode <= 4 && p >= 0.f && p < 1.f && (p == 0.f || seed));
    FW_CHECK_ARG((mode != 1 || res) && (mode != 3 || aux) && (mode != 4 || (aux && period > 0)));
    const unsigned thresh = p > 0.f ? fw_drop_thresh(p) : 0u;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    if (dtype == FW_DT_BF16 && (mode == 2 || mode == 3))
        hipLaunchKernelGGL((dropout_kernel<bf16raw>), dim3(grid_for(n)), dim3(TPB), 0, ST, mode, x, aux, res, y, n, period, (const unsigned*)seed,
                           (unsigned)site, thresh, ik);
    else
        hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid_for(n)), dim3(TPB), 0, ST, mode, x, aux, res, y, n, period, (const unsigned*)seed,
                           (unsigned)site, thresh, ik);
    FW_LAUNCH_RET();
}
extern "C" int fw_rng_tick(void* seed, void* stream) {
    FW_CHECK_ARG(seed);
    hipLaunchKernelGGL(rng_tick_kernel, dim3(1), dim3(64), 0, ST, (unsigned*)seed);
    FW_LAUNCH_RET();
}
extern "C" int fw_add_bcast(const float* x, const float* p, float* out, long n, long period, void* stream) {
    FW_CHECK_ARG(x && p && out && n > 0 && period > 0);
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(n)), dim3(TPB), 0, ST, x, p, out, n, period);
    FW_LAUNCH_RET();
}
// fea: T [B][ED][P];  inter: f32 [B][ED][P];  gap: f32 [B][ED];  mr: f32 [2][ED] (mean, rstd) saved for the backward pass
extern "C" int fw_bn_planes_fwd(int dtype, const void* fea, const float* gamma, const float* beta, float* rmean, float* rvar, long long* nbt,
                                float* mr, float* inter, float* gap, int B, int ED, int P, int training, float eps, float momentum, float slope,
                                void* stream) {
    FW_CHECK_ARG(fea && gamma && beta && rmean && rvar && mr && inter && gap && B > 0 && ED > 0 && P > 0);
    if (dtype == FW_DT_BF16)
        hipLaunchKernelGGL((bn_planes_fwd_kernel<bf16raw>), dim3(ED), dim3(1024), 0, ST, (const bf16raw*)fea, gamma, beta, rmean, rvar, nbt, mr, inter, gap,
                           B, ED, P, training, eps, momentum, slope);
    else
        hipLaunchKernelGGL((bn_planes_fwd_kernel<float>), dim3(ED), dim3(1024), 0, ST, (const float*)fea, gamma, beta, rmean, rvar, nbt, mr, inter, gap, B,
                           ED, P, training, eps, momentum, slope);
    FW_LAUNCH_RET();
}
extern "C" int fw_bn_planes_bwd(int dtype, const void* fea, const float* inter, const float* gamma, const float* mr, const float* dinter,
                                const float* dgap, void* dfea, float* dgamma, float* dbeta, int B, int ED, int P, int training, float slope,
                                void* stream) {
    FW_CHECK_ARG(fea && inter && gamma && mr && dfea && dgamma && dbeta && (dinter || dgap) && B > 0 && ED > 0 && P > 0);
    if (dtype == FW_DT_BF16)
        hipLaunchKernelGGL((bn_planes_bwd_kernel<bf16raw>), dim3(ED), dim3(1024), 0, ST, (const bf16raw*)fea, inter, gamma, mr, dinter, dgap, (bf16raw*)dfea,
                           dgamma, dbeta, B, ED, P, training, slope);
    else
        hipLaunchKernelGGL((bn_planes_bwd_kernel<float>), dim3(ED), dim3(1024), 0, ST, (const float*)fea, inter, gamma, mr, dinter, dgap, (float*)dfea, dgamma,
                           dbeta, B, ED, P, training, slope);
    FW_LAUNCH_RET();
}
extern "C" int fw_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, float slope, void* stream) {
    FW_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0);
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(grid_for((long)M * N)), dim3(TPB), 0, ST, x, w, b, y, M, N, K, slope);
    FW_LAUNCH_RET();
}
extern "C" int fw_small_linear_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, int M, int N,
                                   int K, float slope, void* stream) {
    FW_CHECK_ARG(dy && y && x && w && dx && dw && db && M > 0 && N > 0 && K > 0);
    hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(grid_for((long)M * K + (long)N * K + N)), dim3(TPB), 0, ST, dy, y, x, w, dx, dw, db, M, N, K, slope);
    FW_LAUNCH_RET();
}
