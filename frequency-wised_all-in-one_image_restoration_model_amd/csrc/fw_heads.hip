// Small-tensor kernels of the AirNet hot path:
//   * image band decomposition by partial 2-D DFT (K6; net/utils/frequency_decompose.py:28-118)
//   * encoder contrastive head: BatchNorm2d(batch stats) + LeakyReLU(0.1) + global average pool,
//     forward and backward (K8; net/encoder_Uformer.py:945-951,978-984)
//   * MoCo logits / enqueue (K9; net/utils/moco.py:127-164)
//   * the learned-frequency-selection lambda heads of ALL decoder blocks in one launch
//     (K12; net/decoder_Uformer.py:178-193,279-284)
// These are latency/bandwidth trivia next to the GEMMs; they exist so that no arithmetic of the
// training step is left to a library.
#include "fw_common.h"

namespace {

template <int N> FW_DEV float block_sum(float v, float* red) {      // N threads, N multiple of 64
    v = wave_sum(v);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < N / 64; ++i) s += red[i];
    return s;
}

// =====================================================================================================
// 2-D DFT band decomposition, N <= 256 (power of two)
// =====================================================================================================
// spectrum (unshifted) F[u][v] = sum_{y,x} img[y][x] exp(-2 pi i (u y + v x) / N)  ->  fr, fi [n][N][N]
// One workgroup = 16 output rows of one image (grid: image x row slice): 8x the parallelism of one image per workgroup, and
// no redundant work because each slice contracts over y FIRST:  F[u][v] = sum_x ( sum_y img[y][x] w^{uy} ) w^{xv},  w = e^{-2 pi i/N}.
constexpr int DFT_SL = 16, DFT_MAXN = 256;
__global__ __launch_bounds__(256) void dft2_fwd_kernel(const float* __restrict__ img, float* __restrict__ fr, float* __restrict__ fi, int N) {
    __shared__ float hr[DFT_SL * DFT_MAXN], hi[DFT_SL * DFT_MAXN], tc[DFT_MAXN], ts[DFT_MAXN];
    const int rows = N < DFT_SL ? N : DFT_SL;
    const int u0 = blockIdx.y * rows;
    const float* x = img + (size_t)blockIdx.x * N * N;
    for (int k = threadIdx.x; k < N; k += 256) { float sn, cs; sincospif(2.0f * k / N, &sn, &cs); tc[k] = cs; ts[k] = sn; }
    __syncthreads();
    for (int o = threadIdx.x; o < rows * N; o += 256) {       // H[u][x] = sum_y img[y][x] e^{-i 2pi y u/N}
        const int u = u0 + o / N, xx = o % N;
        float ar = 0.f, ai = 0.f;
        for (int y = 0; y < N; ++y) {
            const float p = x[y * N + xx];
            const int k = (y * u) & (N - 1);
            ar += p * tc[k]; ai -= p * ts[k];
        }
        hr[o] = ar; hi[o] = ai;
    }
    __syncthreads();
    float* outr = fr + (size_t)blockIdx.x * N * N + (size_t)u0 * N;
    float* outi = fi + (size_t)blockIdx.x * N * N + (size_t)u0 * N;
    for (int o = threadIdx.x; o < rows * N; o += 256) {       // F[u][v] = sum_x H[u][x] e^{-i 2pi x v/N}
        const int ul = o / N, v = o % N;
        float ar = 0.f, ai = 0.f;
        for (int xx = 0; xx < N; ++xx) {
            const int k = (xx * v) & (N - 1);
            const float c = tc[k], sn = ts[k], a = hr[ul * N + xx], b = hi[ul * N + xx];
            ar += a * c + b * sn; ai += b * c - a * sn;
        }
        outr[o] = ar; outi[o] = ai;
    }
}
// band image: out[band][n][y][x] = Re( IDFT2( mask_band * F ) ),  mask in UNSHIFTED coordinates [nb][N][N].
// Same slicing (grid: image x band x 16-row slice of y):  out[y][x] = Re sum_v ( sum_u M F[u][v] w^{-uy} ) w^{-vx} / N^2.
__global__ __launch_bounds__(256) void dft2_band_inv_kernel(const float* __restrict__ fr, const float* __restrict__ fi, const float* __restrict__ mask,
                                                            float* __restrict__ out, int N, int nimg) {
    __shared__ float kr[DFT_SL * DFT_MAXN], ki[DFT_SL * DFT_MAXN], tc[DFT_MAXN], ts[DFT_MAXN];
    __shared__ int act[DFT_MAXN];                             // column v holds a non-zero of this band's mask
    const int rows = N < DFT_SL ? N : DFT_SL;
    const int n = blockIdx.x, band = blockIdx.y, y0 = blockIdx.z * rows;
    const float* Fr = fr + (size_t)n * N * N; const float* Fi = fi + (size_t)n * N * N;
    const float* M = mask + (size_t)band * N * N;
    for (int k = threadIdx.x; k < N; k += 256) { float sn, cs; sincospif(2.0f * k / N, &sn, &cs); tc[k] = cs; ts[k] = sn; }
    __syncthreads();
    for (int o = threadIdx.x; o < rows * N; o += 256) {       // K[y][v] = sum_u M F[u][v] e^{+i 2pi u y/N}
        const int y = y0 + o / N, v = o % N;
        float ar = 0.f, ai = 0.f;
        int any = 0;
        for (int u = 0; u < N; ++u) {
            const float m = M[u * N + v];
            if (m == 0.f) continue;
            any = 1;
            const int k = (u * y) & (N - 1);
            const float c = tc[k], sn = ts[k], a = Fr[u * N + v] * m, b = Fi[u * N + v] * m;
            ar += a * c - b * sn; ai += a * sn + b * c;
        }
        kr[o] = ar; ki[o] = ai;
        if (o < N) act[v] = any;                                // first row of the slice: one writer per column
    }
    __syncthreads();
    float* o_ = out + ((size_t)band * nimg + n) * N * N + (size_t)y0 * N;
    const float inv = 1.0f / (N * N);
    for (int o = threadIdx.x; o < rows * N; o += 256) {       // out[y][x] = Re sum_v K[y][v] e^{+i 2pi v x/N} / N^2
        const int yl = o / N, xx = o % N;
        float ar = 0.f;
        for (int v = 0; v < N; ++v) {
            if (!act[v]) continue;                              // uniform over the workgroup: an empty mask column contributes K = 0
            const int k = (v * xx) & (N - 1);
            ar += kr[yl * N + v] * tc[k] - ki[yl * N + v] * ts[k];
        }
        o_[o] = ar * inv;
    }
}
// masked spectrum: mode 0 -> (re, im) interleaved, unshifted (inverse == False);  mode 1 -> |.| in fftshift-ed coordinates ('visual')
__global__ void dft2_band_spec_kernel(const float* __restrict__ fr, const float* __restrict__ fi, const float* __restrict__ mask,
                                      float* __restrict__ out, int N, int nimg, int mode) {
    const int n = blockIdx.x, band = blockIdx.y;
    const float* Fr = fr + (size_t)n * N * N; const float* Fi = fi + (size_t)n * N * N;
    const float* M = mask + (size_t)band * N * N;
    for (int o = threadIdx.x; o < N * N; o += 256) {
        const float m = M[o], a = Fr[o] * m, b = Fi[o] * m;
        if (mode == 0) {
            float* d = out + (((size_t)band * nimg + n) * N * N + o) * 2;
            d[0] = a; d[1] = b;
        } else {
            const int u = o / N, v = o % N;
            const int us = (u + N / 2) & (N - 1), vs = (v + N / 2) & (N - 1);
            out[((size_t)band * nimg + n) * N * N + us * N + vs] = sqrtf(a * a + b * b);
        }
    }
}
// last band of a decomposition whose masks partition the spectrum: out[nb-1] = img - sum_{b < nb-1} out[b]   (sum_b M_b = 1)
__global__ __launch_bounds__(256) void band_residual_kernel(const float* __restrict__ img, float* __restrict__ out, long per_band, int nbands) {
    const long i = (blockIdx.x * 256L + threadIdx.x) * 4;
    if (i >= per_band) return;
    f32x4 r = *reinterpret_cast<const f32x4*>(img + i);
    for (int b = 0; b + 1 < nbands; ++b) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(out + b * per_band + i);
        r[0] -= t[0]; r[1] -= t[1]; r[2] -= t[2]; r[3] -= t[3];
    }
    *reinterpret_cast<f32x4*>(out + (nbands - 1) * per_band + i) = r;
}
// DC split (frequency_decompose_dc): out[0] = mean, out[1] = x - mean
__global__ __launch_bounds__(256) void dc_split_kernel(const float* __restrict__ img, float* __restrict__ out, int NN, int nimg) {
    __shared__ float red[4];
    const float* x = img + (size_t)blockIdx.x * NN;
    float s = 0.f;
    for (int o = threadIdx.x; o < NN; o += 256) s += x[o];
    const float mean = block_sum<256>(s, red) / NN;
    float* o0 = out + (size_t)blockIdx.x * NN; float* o1 = out + ((size_t)nimg + blockIdx.x) * NN;
    for (int o = threadIdx.x; o < NN; o += 256) { o0[o] = mean; o1[o] = x[o] - mean; }
}

// =====================================================================================================
// The same decomposition as ONE kernel on the f32 MFMA (N = 64, 128), for band masks that PARTITION the spectrum (the encoder's
// `frequency_decompose_1` pre-processing, encoder_Uformer.py:964-966; frequency_decompose.py:70-107):
//     out[b] = Re IDFT2( M_b . DFT2(x) )  for b < nb - 1,      out[nb-1] = x - sum_{b < nb-1} out[b]
// One workgroup (N / 16 waves) per image; every transform is a chain of 128^3 real products v_mfma_f32_16x16x4_f32 (exact f32
// products, f32 accumulation -- the arithmetic class of the scalar kernels above):
//     T = x^T (C - iS)        wave w: columns kappa of x (its 16), all frequencies v        -> LDS  Ts[kappa][v]
//     F = (C - iS) T          wave w: frequencies v (its 16), all u; Ts read k-major        -> registers (kept for every band)
//   per band:  Y = M_b . F -> LDS Ys[v][u] (own rows);  Z = (C + iS) Y  (wave-local) -> LDS Zs[v][kappa];
//              out[rho][kappa] = Re( Z (C + iS) ) / N^2     wave w: image columns kappa (its 16), all rows rho
// cos / sin / -sin panels [N][N] are read as ready-made MFMA fragments from L2 (192 KB, shared by every workgroup); LDS holds one
// complex N x N f32 matrix (135 KB at N = 128).  A band whose mask is the DC bin alone is the image mean (no transform).
// Replaces dft2_fwd + dft2_band_inv + band_residual: 497 us -> ~100 us per call at 48 x 128 x 128.
// =====================================================================================================
FW_DEV uint4 dftp_frag(const float* P, int N, int row0, int c) {          // A-operand fragment of a global f32 [N][N] panel
    const int l = lane_id();
    return *reinterpret_cast<const uint4*>(P + (size_t)(row0 + (l & 15)) * N + c * 16 + ((l >> 4) << 2));
}
template <int N>
__global__ __launch_bounds__(N * 4) void dft2_decompose_mfma_kernel(const float* __restrict__ img, const float* __restrict__ mask, const float* __restrict__ panels,
                                                                     float* __restrict__ out, int nimg, int nbands, unsigned dc_bits) {
    constexpr int MT = N / 16, KC = N / 16, LD = N * 4 + 16, BUF = N * LD;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    __shared__ float mean_s;
    char* Br = dsm; char* Bi = dsm + BUF;
    const float* Cg = panels; const float* Sg = panels + N * N; const float* Ng = panels + 2 * N * N;
    const int w = threadIdx.x >> 6, l = lane_id();
    const size_t n = blockIdx.x;
    const float* x = img + n * N * N;
    const float inv = 1.0f / (float)(N * N);
    f32x4 a1[MT], a2[MT];
    // ---- T[kappa][v] = sum_rho x[rho][kappa] (C - iS)[rho][v]:  acc(m = v, n = kappa own strip); B fragments straight from the image
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { a1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[mt] = a1[mt]; }
#pragma unroll 2
    for (int c = 0; c < KC; ++c) {
        const float* xp = x + (size_t)(c * 16 + ((l >> 4) << 2)) * N + 16 * w + (l & 15);
        const uint4 bf = make_uint4(__float_as_uint(xp[0]), __float_as_uint(xp[N]), __float_as_uint(xp[2 * N]), __float_as_uint(xp[3 * N]));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            mma_chunk<float>(a1[mt], dftp_frag(Cg, N, 16 * mt, c), bf);
            mma_chunk<float>(a2[mt], dftp_frag(Ng, N, 16 * mt, c), bf);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        store_acc_T<float>(Br + 16 * w * LD, LD, 16 * mt, 0, a1[mt]);          // Ts[kappa][v], v contiguous, rows kappa = own strip
        store_acc_T<float>(Bi + 16 * w * LD, LD, 16 * mt, 0, a2[mt]);
    }
    __syncthreads();
    // ---- F[u][v] = sum_kappa (C - iS)[u][kappa] T[kappa][v]:  acc(m = u, n = v own strip); Ts read k-major
    f32x4 fr[MT], fi[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { fr[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; fi[mt] = fr[mt]; }
#pragma unroll 2
    for (int c = 0; c < KC; ++c) {
        const uint4 br = frag_km<float>(Br, LD, 16 * w, c), bi = frag_km<float>(Bi, LD, 16 * w, c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint4 cf = dftp_frag(Cg, N, 16 * mt, c);
            mma_chunk<float>(fr[mt], cf, br);
            mma_chunk<float>(fr[mt], dftp_frag(Sg, N, 16 * mt, c), bi);
            mma_chunk<float>(fi[mt], cf, bi);
            mma_chunk<float>(fi[mt], dftp_frag(Ng, N, 16 * mt, c), br);
        }
    }
    if (threadIdx.x == 0) mean_s = fr[0][0] * inv;                             // F[0][0] / N^2 (wave 0, lane 0, tile 0, row 0)
    // the last band starts as x and loses every other band: rows rho = 16 mt + 4 (l >> 4) + r, column kappa = 16 w + (l & 15)
    f32x4 rest[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rest[mt][r] = x[(size_t)(16 * mt + 4 * (l >> 4) + r) * N + 16 * w + (l & 15)];
    for (int b = 0; b + 1 < nbands; ++b) {
        float* ob = out + ((size_t)b * nimg + n) * N * N;
        __syncthreads();                                                       // everybody has left the buffers (T, or the previous band's Z)
        if ((dc_bits >> b) & 1u) {
            const float m = mean_s;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ob[(size_t)(16 * mt + 4 * (l >> 4) + r) * N + 16 * w + (l & 15)] = m; rest[mt][r] -= m; }
            continue;
        }
        const float* M = mask + (size_t)b * N * N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {                                       // Ys[v][u] = M . F, u contiguous, rows v = own strip
            f32x4 mk;
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[r] = M[(size_t)(16 * mt + 4 * (l >> 4) + r) * N + 16 * w + (l & 15)];
            store_acc_T<float>(Br + 16 * w * LD, LD, 16 * mt, 0, fr[mt] * mk);
            store_acc_T<float>(Bi + 16 * w * LD, LD, 16 * mt, 0, fi[mt] * mk);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- Z[kappa][v] = sum_u (C + iS)[kappa][u] Y[u][v]:  acc(m = kappa, n = v own strip), wave-local operands
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { a1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[mt] = a1[mt]; }
#pragma unroll 2
        for (int c = 0; c < KC; ++c) {
            const uint4 br = frag_kc(Br + 16 * w * LD, LD, 0, c), bi = frag_kc(Bi + 16 * w * LD, LD, 0, c);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint4 cf = dftp_frag(Cg, N, 16 * mt, c);
                mma_chunk<float>(a1[mt], cf, br);
                mma_chunk<float>(a1[mt], dftp_frag(Ng, N, 16 * mt, c), bi);
                mma_chunk<float>(a2[mt], cf, bi);
                mma_chunk<float>(a2[mt], dftp_frag(Sg, N, 16 * mt, c), br);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {                                       // Zs[v][kappa], kappa contiguous, rows v = own strip
            store_acc_T<float>(Br + 16 * w * LD, LD, 16 * mt, 0, a1[mt]);
            store_acc_T<float>(Bi + 16 * w * LD, LD, 16 * mt, 0, a2[mt]);
        }
        __syncthreads();
        // ---- out[rho][kappa] = Re sum_v Z[kappa][v] (C + iS)[v][rho] / N^2:  acc(m = rho, n = kappa own strip); Zs read k-major
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int c = 0; c < KC; ++c) {
            const uint4 br = frag_km<float>(Br, LD, 16 * w, c), bi = frag_km<float>(Bi, LD, 16 * w, c);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                mma_chunk<float>(a1[mt], dftp_frag(Cg, N, 16 * mt, c), br);
                mma_chunk<float>(a1[mt], dftp_frag(Ng, N, 16 * mt, c), bi);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = a1[mt][r] * inv;
                ob[(size_t)(16 * mt + 4 * (l >> 4) + r) * N + 16 * w + (l & 15)] = v;
                rest[mt][r] -= v;
            }
    }
    float* ol = out + ((size_t)(nbands - 1) * nimg + n) * N * N;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ol[(size_t)(16 * mt + 4 * (l >> 4) + r) * N + 16 * w + (l & 15)] = rest[mt][r];
}

// =====================================================================================================
// encoder head: BatchNorm2d + LeakyReLU(0.1) + global average pool over fea viewed as [B][ED][P]
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ fea, float* __restrict__ part, int B, int ED, int P) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const T* x = fea + ((size_t)b * ED + c) * P;
    // SHIFTED sums: deviations from a pivot of the channel (its first element in image 0), so that var = E[d^2] - E[d]^2 does not
    // cancel when the plane's spread is small against its mean (f32: the un-shifted form lost 3 digits of the variance on the seeded
    // fixtures, 0.4 % in the 65 536-wide head's weight gradient -- tests/test_engine_parity_gpu.py, golden model_all3_kdiff)
    const float pivot = TT<T>::ld(fea + (size_t)c * P);
    float s = 0.f, q = 0.f;
    constexpr int E = TT<T>::E16;
    if (P % (2 * E * 256) == 0 && ((uintptr_t)x & 15) == 0) {          // 16-byte loads, two in flight per lane (the scalar form: 1.6 TB/s on the 65 536-wide plane)
        float s1 = 0.f, q1 = 0.f;
        for (int o = threadIdx.x * E; o < P; o += 2 * E * 256) {
            const uint4 ra = *reinterpret_cast<const uint4*>(x + o), rb = *reinterpret_cast<const uint4*>(x + o + E * 256);
            float fa[E], fb[E];
            unpack16<T>(ra, fa); unpack16<T>(rb, fb);
#pragma unroll
            for (int e = 0; e < E; ++e) { const float va = fa[e] - pivot, vb = fb[e] - pivot; s += va; q += va * va; s1 += vb; q1 += vb * vb; }
        }
        s += s1; q += q1;
    } else {
        for (int o = threadIdx.x; o < P; o += 256) { const float v = TT<T>::ld(x + o) - pivot; s += v; q += v * v; }
    }
    s = block_sum<256>(s, red); q = block_sum<256>(q, red);
    if (threadIdx.x == 0) { part[((size_t)c * B + b) * 2] = s; part[((size_t)c * B + b) * 2 + 1] = q; }
}
// training: stats from `part`; eval: running stats.  gap[b][c] = mean_p lrelu(bn(x)).  Block (c, 0) updates the running stats.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_gap_kernel(const T* __restrict__ fea, const float* __restrict__ part, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                           long long* __restrict__ nbt, float* __restrict__ gap, float* __restrict__ saved, int B,
                                                           int ED, int P, int training, float eps, float mom, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    float mean, var;
    if (training) {
        float s = 0.f, q = 0.f;
        for (int i = 0; i < B; ++i) { s += part[((size_t)c * B + i) * 2]; q += part[((size_t)c * B + i) * 2 + 1]; }
        const float n = (float)B * P;
        const float md = s / n;                                    // mean of the deviations from the pivot (bn_stats_kernel)
        mean = TT<T>::ld(fea + (size_t)c * P) + md; var = fmaxf(q / n - md * md, 0.f);
        if (b == 0 && threadIdx.x == 0) {
            rmean[c] = (1.f - mom) * rmean[c] + mom * mean;
            rvar[c] = (1.f - mom) * rvar[c] + mom * var * n / (n - 1.f);
            if (c == 0) *nbt += 1;
            saved[c * 2] = mean; saved[c * 2 + 1] = rsqrtf(var + eps);
        }
    } else { mean = rmean[c]; var = rvar[c]; }
    const float rs = rsqrtf(var + eps), g = gamma[c], be = beta[c];
    const T* x = fea + ((size_t)b * ED + c) * P;
    float s = 0.f;
    constexpr int E = TT<T>::E16;
    if (P % (2 * E * 256) == 0 && ((uintptr_t)x & 15) == 0) {
        float s1 = 0.f;
        for (int o = threadIdx.x * E; o < P; o += 2 * E * 256) {
            const uint4 ra = *reinterpret_cast<const uint4*>(x + o), rb = *reinterpret_cast<const uint4*>(x + o + E * 256);
            float fa[E], fb[E];
            unpack16<T>(ra, fa); unpack16<T>(rb, fb);
#pragma unroll
            for (int e = 0; e < E; ++e) { s += lrelu_f((fa[e] - mean) * rs * g + be, slope); s1 += lrelu_f((fb[e] - mean) * rs * g + be, slope); }
        }
        s += s1;
    } else {
        for (int o = threadIdx.x; o < P; o += 256) s += lrelu_f((TT<T>::ld(x + o) - mean) * rs * g + be, slope);
    }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) gap[(size_t)b * ED + c] = s / P;
}
// backward pass 1: dy = dgap[b][c]/P * lrelu'(z); part2[c][b] = (sum dy, sum dy*xhat)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ fea, const float* __restrict__ saved, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ dgap, float* __restrict__ part2,
                                                           int B, int ED, int P, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y;
    const float mean = saved[c * 2], rs = saved[c * 2 + 1], g = gamma[c], be = beta[c];
    const float dg = dgap[(size_t)b * ED + c] / P;
    const T* x = fea + ((size_t)b * ED + c) * P;
    float s = 0.f, q = 0.f;
    constexpr int E = TT<T>::E16;
    if (P % (2 * E * 256) == 0 && ((uintptr_t)x & 15) == 0) {
        float s1 = 0.f, q1 = 0.f;
        for (int o = threadIdx.x * E; o < P; o += 2 * E * 256) {
            const uint4 ra = *reinterpret_cast<const uint4*>(x + o), rb = *reinterpret_cast<const uint4*>(x + o + E * 256);
            float fa[E], fb[E];
            unpack16<T>(ra, fa); unpack16<T>(rb, fb);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float xa = (fa[e] - mean) * rs, xb = (fb[e] - mean) * rs;
                const float da = (xa * g + be > 0.f) ? dg : dg * slope, db_ = (xb * g + be > 0.f) ? dg : dg * slope;
                s += da; q += da * xa; s1 += db_; q1 += db_ * xb;
            }
        }
        s += s1; q += q1;
    } else {
        for (int o = threadIdx.x; o < P; o += 256) {
            const float xh = (TT<T>::ld(x + o) - mean) * rs;
            const float dy = (xh * g + be > 0.f) ? dg : dg * slope;
            s += dy; q += dy * xh;
        }
    }
    s = block_sum<256>(s, red); q = block_sum<256>(q, red);
    if (threadIdx.x == 0) { part2[((size_t)c * B + b) * 2] = s; part2[((size_t)c * B + b) * 2 + 1] = q; }
}
// backward pass 2: dx = g*rs*(dy - mean(dy) - xhat*mean(dy*xhat)) -> T;  dgamma/dbeta += (block b == 0)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ fea, const float* __restrict__ saved, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ dgap, const float* __restrict__ part2,
                                                           T* __restrict__ dfea, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int ED,
                                                           int P, float slope) {
    const int c = blockIdx.x, b = blockIdx.y;
    const float mean = saved[c * 2], rs = saved[c * 2 + 1], g = gamma[c], be = beta[c];
    float s = 0.f, q = 0.f;
    for (int i = 0; i < B; ++i) { s += part2[((size_t)c * B + i) * 2]; q += part2[((size_t)c * B + i) * 2 + 1]; }
    if (b == 0 && threadIdx.x == 0) { atomicAdd(dbeta + c, s); atomicAdd(dgamma + c, q); }
    const float n = (float)B * P, m1 = s / n, m2 = q / n;
    const float dg = dgap[(size_t)b * ED + c] / P;
    const T* x = fea + ((size_t)b * ED + c) * P;
    T* d = dfea + ((size_t)b * ED + c) * P;
    for (int o = threadIdx.x; o < P; o += 256) {
        const float xh = (TT<T>::ld(x + o) - mean) * rs;
        const float dy = (xh * g + be > 0.f) ? dg : dg * slope;
        TT<T>::st(d + o, g * rs * (dy - m1 - xh * m2));
    }
}

// =====================================================================================================
// MoCo: normalise, logits = [q.k, q.queue] / T, backward to q, enqueue
// =====================================================================================================
// grid (L, B), 256 threads.  q, k: [L][B][ED] (un-normalised).  queue: [L][ED][K].  logits: [L][B][1+K].  khat: [L][B][ED].
__global__ __launch_bounds__(256) void moco_logits_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ queue,
                                                          float* __restrict__ logits, float* __restrict__ khat, int B, int ED, int K, float invT) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float sm[];      // qh[ED]
    const int l = blockIdx.x, b = blockIdx.y;
    const float* qq = q + ((size_t)l * B + b) * ED; const float* kk = k + ((size_t)l * B + b) * ED;
    float sq = 0.f, sk = 0.f;
    for (int c = threadIdx.x; c < ED; c += 256) { sq += qq[c] * qq[c]; sk += kk[c] * kk[c]; }
    sq = block_sum<256>(sq, red); sk = block_sum<256>(sk, red);
    const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
    float pos = 0.f;
    for (int c = threadIdx.x; c < ED; c += 256) {
        const float a = qq[c] * iq, bb = kk[c] * ik;
        sm[c] = a; khat[((size_t)l * B + b) * ED + c] = bb; pos += a * bb;
    }
    pos = block_sum<256>(pos, red);
    float* lg = logits + ((size_t)l * B + b) * (1 + K);
    if (threadIdx.x == 0) lg[0] = pos * invT;
    const float* Q = queue + (size_t)l * ED * K;
    for (int j = threadIdx.x; j < K; j += 256) {
        float s = 0.f;
        for (int c = 0; c < ED; ++c) s += sm[c] * Q[(size_t)c * K + j];
        lg[1 + j] = s * invT;
    }
}
// dq[l][b][c] from dlogits; queue = the queue the logits were computed with
__global__ __launch_bounds__(256) void moco_logits_bwd_kernel(const float* __restrict__ q, const float* __restrict__ khat, const float* __restrict__ queue,
                                                              const float* __restrict__ dlogits, float* __restrict__ dq, int B, int ED, int K, float invT) {
    __shared__ float red[4];
    const int l = blockIdx.x, b = blockIdx.y;
    const float* qq = q + ((size_t)l * B + b) * ED; const float* kh = khat + ((size_t)l * B + b) * ED;
    const float* dl = dlogits + ((size_t)l * B + b) * (1 + K);
    const float* Q = queue + (size_t)l * ED * K;
    float sq = 0.f;
    for (int c = threadIdx.x; c < ED; c += 256) sq += qq[c] * qq[c];
    sq = block_sum<256>(sq, red);
    const float nq = fmaxf(sqrtf(sq), 1e-12f), iq = 1.f / nq;
    // dqh[c] = (dl0 * kh[c] + sum_j dl[1+j] Q[c][j]) / T ;  dq = (dqh - qh (qh . dqh)) / |q|
    float dot = 0.f;
    float loc[4];                                   // ED <= 1024
    for (int t = 0; t < 4; ++t) {
        const int c = threadIdx.x + 256 * t;
        loc[t] = 0.f;
        if (c < ED) {
            float s = dl[0] * kh[c];
            for (int j = 0; j < K; ++j) s += dl[1 + j] * Q[(size_t)c * K + j];
            loc[t] = s * invT;
            dot += loc[t] * qq[c] * iq;
        }
    }
    dot = block_sum<256>(dot, red);
    for (int t = 0; t < 4; ++t) {
        const int c = threadIdx.x + 256 * t;
        if (c < ED) dq[((size_t)l * B + b) * ED + c] = (loc[t] - qq[c] * iq * dot) * iq;
    }
}
// queue[l][c][ptr + b] = khat[l][b][c];  ptr = (ptr + B) % K   (moco.py:52-66)
__global__ void moco_enqueue_kernel(float* __restrict__ queue, const float* __restrict__ khat, long long* __restrict__ ptr, int L, int B, int ED, int K) {
    const int p = (int)*ptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L * B * ED; i += gridDim.x * blockDim.x) {
        const int c = i % ED, b = (i / ED) % B, l = i / (ED * B);
        queue[((size_t)l * ED + c) * K + p + b] = khat[i];
    }
}
__global__ void moco_ptr_kernel(long long* __restrict__ ptr, int B, int K) { *ptr = (*ptr + B) % K; }

// =====================================================================================================
// LFS lambda heads
// =====================================================================================================
// xbar[i][b][c] = mean_t xhat[t][c], xhat = LayerNorm-normalised (no affine) rows of inter[i][b] ([NT tokens][C]; NT = (S/16)^2: 64 at 128x128)
constexpr int LFS_MAXT = 1024;
__global__ __launch_bounds__(256) void lfs_xbar_kernel(const float* __restrict__ inter, float* __restrict__ xbar, float* __restrict__ stats, int NT, int C, float eps) {
    __shared__ float mu[LFS_MAXT], rs[LFS_MAXT];
    const size_t blk = (size_t)blockIdx.x;                 // (band, b) flattened
    const float* x = inter + blk * NT * C;
    const int w = threadIdx.x >> 6, l = lane_id();
    for (int t = w; t < NT; t += 4) {
        float s = 0.f;
        for (int c = l; c < C; c += 64) s += x[(size_t)t * C + c];
        const float m = wave_sum(s) / C;
        float q = 0.f;
        for (int c = l; c < C; c += 64) { const float d = x[(size_t)t * C + c] - m; q += d * d; }
        const float r = rsqrtf(wave_sum(q) / C + eps);
        if (l == 0) { mu[t] = m; rs[t] = r; stats[(blk * NT + t) * 2] = m; stats[(blk * NT + t) * 2 + 1] = r; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int t = 0; t < NT; ++t) s += (x[(size_t)t * C + c] - mu[t]) * rs[t];
        xbar[blk * C + c] = s / NT;
    }
}
// dinter[i][b][t][c] += LN-backward of dxhat[t][c] = dxbar[c] / NT
__global__ __launch_bounds__(256) void lfs_xbar_bwd_kernel(const float* __restrict__ inter, const float* __restrict__ stats, const float* __restrict__ dxbar,
                                                           float* __restrict__ dinter, int NT, int C) {
    __shared__ float red[4];
    const size_t blk = (size_t)blockIdx.x;
    const float* x = inter + blk * NT * C; float* dx = dinter + blk * NT * C;
    const float* g = dxbar + blk * C;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += g[c];
    const float m1 = block_sum<256>(s, red) / C / NT;
    const int w = threadIdx.x >> 6, l = lane_id();
    for (int t = w; t < NT; t += 4) {
        const float m = stats[(blk * NT + t) * 2], r = stats[(blk * NT + t) * 2 + 1];
        float q = 0.f;
        for (int c = l; c < C; c += 64) q += g[c] / NT * (x[(size_t)t * C + c] - m) * r;
        const float m2 = wave_sum(q) / C;
        for (int c = l; c < C; c += 64) {
            const float xh = (x[(size_t)t * C + c] - m) * r;
            dx[(size_t)t * C + c] += r * (g[c] / NT - m1 - xh * m2);
        }
    }
}

// pointer table per (block, band): 0 ln.w 1 ln.b 2 lin.w[h][C] 3 lin.b 4 mlp0.w[h][h] 5 mlp0.b 6 mlp2.w 7 mlp2.b
// grid (nblk, B), 64 threads.  coef at coef + coef_off[blk] as [B][h][3].  save: [nblk][2][B][16][3] (u, a1, s)
__global__ __launch_bounds__(64) void lfs_lambda_kernel(const float* __restrict__ xbar, const unsigned long long* __restrict__ ptab,
                                                        const int* __restrict__ heads, const long long* __restrict__ coef_off,
                                                        float* __restrict__ coef, float* __restrict__ save, int B, int C, int nb1) {
    __shared__ float u[16], s1[16], lam[2][16];
    const int blk = blockIdx.x, b = blockIdx.y, h = heads[blk], l = lane_id();
    for (int band = 0; band < nb1; ++band) {
        const unsigned long long* pt = ptab + ((size_t)blk * 2 + band) * 8;
        const float* lnw = (const float*)pt[0]; const float* lnb = (const float*)pt[1];
        const float* W = (const float*)pt[2]; const float* bl = (const float*)pt[3];
        const float* W1 = (const float*)pt[4]; const float* b1 = (const float*)pt[5];
        const float* W2 = (const float*)pt[6]; const float* b2 = (const float*)pt[7];
        const float* xb = xbar + ((size_t)band * B + b) * C;
        for (int hh = 0; hh < h; ++hh) {
            float s = 0.f;
            for (int c = l; c < C; c += 64) s += W[(size_t)hh * C + c] * (lnw[c] * xb[c] + lnb[c]);
            s = wave_sum(s);
            if (l == 0) u[hh] = s + bl[hh];
        }
        __syncthreads();
        float* sv = save + ((((size_t)blk * 2 + band) * B + b) * 16) * 3;
        if (l < h) {
            float a = b1[l];
            for (int j = 0; j < h; ++j) a += W1[l * h + j] * u[j];
            s1[l] = lrelu_f(a, 0.1f);
            sv[l * 3] = u[l]; sv[l * 3 + 1] = a; sv[l * 3 + 2] = s1[l];
        }
        __syncthreads();
        if (l < h) {
            float a = b2[l];
            for (int j = 0; j < h; ++j) a += W2[l * h + j] * s1[j];
            lam[band][l] = a;
        }
        __syncthreads();
    }
    if (l < h) {
        float* cf = coef + coef_off[blk] + ((size_t)b * h + l) * 3;
        if (nb1 == 2) { const float l1 = lam[0][l], l2 = lam[1][l]; cf[0] = 1.f + l2; cf[1] = -l2 * (1.f / 64.f); cf[2] = l1 - l2; }
        else { const float l1 = lam[0][l]; cf[0] = 1.f + l1; cf[1] = -l1 * (1.f / 64.f); cf[2] = 0.f; }
    }
}
// gtab: same layout as ptab but pointing at the gradient tensors (accumulated with atomics)
__global__ __launch_bounds__(64) void lfs_lambda_bwd_kernel(const float* __restrict__ xbar, const unsigned long long* __restrict__ ptab,
                                                            const unsigned long long* __restrict__ gtab, const int* __restrict__ heads,
                                                            const long long* __restrict__ coef_off, const float* __restrict__ dcoef,
                                                            const float* __restrict__ save, float* __restrict__ dxbar, int B, int C, int nb1) {
    __shared__ float dl[16], da1[16], du[16];
    const int blk = blockIdx.x, b = blockIdx.y, h = heads[blk], l = lane_id();
    for (int band = 0; band < nb1; ++band) {
        const unsigned long long* pt = ptab + ((size_t)blk * 2 + band) * 8;
        const unsigned long long* gt = gtab + ((size_t)blk * 2 + band) * 8;
        const float* lnw = (const float*)pt[0]; const float* lnb = (const float*)pt[1];
        const float* W = (const float*)pt[2]; const float* W1 = (const float*)pt[4]; const float* W2 = (const float*)pt[6];
        float* g_lnw = (float*)gt[0]; float* g_lnb = (float*)gt[1]; float* gW = (float*)gt[2]; float* gbl = (float*)gt[3];
        float* gW1 = (float*)gt[4]; float* gb1 = (float*)gt[5]; float* gW2 = (float*)gt[6]; float* gb2 = (float*)gt[7];
        const float* xb = xbar + ((size_t)band * B + b) * C;
        const float* sv = save + ((((size_t)blk * 2 + band) * B + b) * 16) * 3;
        if (l < h) {
            const float* dc = dcoef + coef_off[blk] + ((size_t)b * h + l) * 3;
            float d;
            if (nb1 == 2) d = band == 0 ? dc[2] : (dc[0] - dc[1] * (1.f / 64.f) - dc[2]);
            else d = dc[0] - dc[1] * (1.f / 64.f);
            dl[l] = d;
        }
        __syncthreads();
        if (l < h) {
            atomicAdd(gb2 + l, dl[l]);
            for (int j = 0; j < h; ++j) atomicAdd(gW2 + l * h + j, dl[l] * sv[j * 3 + 2]);
            float ds = 0.f;
            for (int j = 0; j < h; ++j) ds += W2[j * h + l] * dl[j];
            da1[l] = sv[l * 3 + 1] > 0.f ? ds : ds * 0.1f;
        }
        __syncthreads();
        if (l < h) {
            atomicAdd(gb1 + l, da1[l]);
            for (int j = 0; j < h; ++j) atomicAdd(gW1 + l * h + j, da1[l] * sv[j * 3]);
            float d = 0.f;
            for (int j = 0; j < h; ++j) d += W1[j * h + l] * da1[j];
            du[l] = d;
            atomicAdd(gbl + l, d);
        }
        __syncthreads();
        for (int c = l; c < C; c += 64) {
            const float z = lnw[c] * xb[c] + lnb[c];
            float dz = 0.f;
            for (int hh = 0; hh < h; ++hh) { atomicAdd(gW + (size_t)hh * C + c, du[hh] * z); dz += W[(size_t)hh * C + c] * du[hh]; }
            atomicAdd(g_lnw + c, dz * xb[c]);
            atomicAdd(g_lnb + c, dz);
            atomicAdd(dxbar + ((size_t)band * B + b) * C + c, dz * lnw[c]);
        }
        __syncthreads();
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

// spectrum scratch fr/fi: [nimg][N][N] f32 each.
extern "C" int fw_dft2_fwd(const float* img, float* fr, float* fi, int nimg, int N, void* stream) {
    FW_CHECK_ARG(img && fr && fi && nimg > 0 && N >= 8 && N <= DFT_MAXN && (N & (N - 1)) == 0);
    hipLaunchKernelGGL(dft2_fwd_kernel, dim3(nimg, N < DFT_SL ? 1 : N / DFT_SL), dim3(256), 0, ST, img, fr, fi, N);
    FW_LAUNCH_RET();
}
// mode 0: real band images (inverse=True) out [nb][nimg][N][N];  1: (re,im) pairs out [nb][nimg][N][N][2];  2: |.| fftshift-ed ('visual')
extern "C" int fw_dft2_bands(const float* fr, const float* fi, const float* mask_unshifted, float* out, int nimg, int N, int nbands,
                             int mode, void* stream) {
    FW_CHECK_ARG(fr && fi && mask_unshifted && out && nimg > 0 && N >= 8 && N <= DFT_MAXN && (N & (N - 1)) == 0 && nbands > 0 && mode >= 0 && mode <= 2);
    if (mode == 0) {
        hipLaunchKernelGGL(dft2_band_inv_kernel, dim3(nimg, nbands, N < DFT_SL ? 1 : N / DFT_SL), dim3(256), 0, ST, fr, fi, mask_unshifted, out, N, nimg);
    } else {
        hipLaunchKernelGGL(dft2_band_spec_kernel, dim3(nimg, nbands), dim3(256), 0, ST, fr, fi, mask_unshifted, out, N, nimg, mode - 1);
    }
    FW_LAUNCH_RET();
}
// The whole decomposition of a PARTITIONING mask set in one launch on the f32 MFMA (N = 64 or 128): out [nbands][nimg][N][N];
// mask: f32 [nbands][N][N] un-shifted; panels: f32 cos | sin | -sin [3][N][N] of 2 pi u i / N; dc_bits: bit b = band b is the DC bin alone
extern "C" int fw_dft2_decompose(const float* img, const float* mask, const float* panels, float* out, int nimg, int N, int nbands, int dc_bits,
                                 void* stream) {
    FW_CHECK_ARG(img && mask && panels && out && nimg > 0 && nbands >= 2 && nbands <= 31 && (N == 64 || N == 128));
    const size_t lds = (size_t)2 * N * (N * 4 + 16);
    if (N == 128) {
        FW_SET_LDS_ONCE(dft2_decompose_mfma_kernel<128>, lds);
        hipLaunchKernelGGL(dft2_decompose_mfma_kernel<128>, dim3(nimg), dim3(512), lds, ST, img, mask, panels, out, nimg, nbands, (unsigned)dc_bits);
    } else {
        FW_SET_LDS_ONCE(dft2_decompose_mfma_kernel<64>, lds);
        hipLaunchKernelGGL(dft2_decompose_mfma_kernel<64>, dim3(nimg), dim3(256), lds, ST, img, mask, panels, out, nimg, nbands, (unsigned)dc_bits);
    }
    FW_LAUNCH_RET();
}
// out: [nbands][nimg][N][N] whose first nbands-1 bands are filled (fw_dft2_bands with nbands-1); img: [nimg][N][N]
extern "C" int fw_band_residual(const float* img, float* out, int nimg, int N, int nbands, void* stream) {
    FW_CHECK_ARG(img && out && nimg > 0 && N >= 8 && N % 4 == 0 && nbands >= 2);
    const long per_band = (long)nimg * N * N;
    hipLaunchKernelGGL(band_residual_kernel, dim3((unsigned)((per_band / 4 + 255) / 256)), dim3(256), 0, ST, img, out, per_band, nbands);
    FW_LAUNCH_RET();
}
extern "C" int fw_dc_split(const float* img, float* out, int nimg, int NN, void* stream) {
    FW_CHECK_ARG(img && out && nimg > 0 && NN > 0);
    hipLaunchKernelGGL(dc_split_kernel, dim3(nimg), dim3(256), 0, ST, img, out, NN, nimg);
    FW_LAUNCH_RET();
}

// fea: T [B][ED][P] (the 448 -> ED*256 Linear output viewed per sample).  part: f32 [ED][B][2] scratch.  saved: f32 [ED][2].
extern "C" int fw_bn_lrelu_gap_fwd(int dtype, const void* fea, const float* gamma, const float* beta, float* rmean, float* rvar,
                                   long long* nbt, float* part, float* saved, float* gap, int B, int ED, int P, int training,
                                   float eps, float momentum, float slope, void* stream) {
    FW_CHECK_ARG(fea && gamma && beta && rmean && rvar && gap && B > 0 && ED > 0 && P > 0 && (!training || (part && saved && nbt)));
    if (dtype == FW_DT_BF16) {
        if (training) hipLaunchKernelGGL((bn_stats_kernel<bf16raw>), dim3(ED, B), dim3(256), 0, ST, (const bf16raw*)fea, part, B, ED, P);
        hipLaunchKernelGGL((bn_apply_gap_kernel<bf16raw>), dim3(ED, B), dim3(256), 0, ST, (const bf16raw*)fea, part, gamma, beta, rmean, rvar, nbt,
                           gap, saved, B, ED, P, training, eps, momentum, slope);
    } else {
        if (training) hipLaunchKernelGGL((bn_stats_kernel<float>), dim3(ED, B), dim3(256), 0, ST, (const float*)fea, part, B, ED, P);
        hipLaunchKernelGGL((bn_apply_gap_kernel<float>), dim3(ED, B), dim3(256), 0, ST, (const float*)fea, part, gamma, beta, rmean, rvar, nbt, gap,
                           saved, B, ED, P, training, eps, momentum, slope);
    }
    FW_LAUNCH_RET();
}
extern "C" int fw_bn_lrelu_gap_bwd(int dtype, const void* fea, const float* gamma, const float* beta, const float* saved, const float* dgap,
                                   float* part2, void* dfea, float* dgamma, float* dbeta, int B, int ED, int P, float slope, void* stream) {
    FW_CHECK_ARG(fea && gamma && beta && saved && dgap && part2 && dfea && dgamma && dbeta);
    if (dtype == FW_DT_BF16) {
        hipLaunchKernelGGL((bn_bwd_stats_kernel<bf16raw>), dim3(ED, B), dim3(256), 0, ST, (const bf16raw*)fea, saved, gamma, beta, dgap, part2, B, ED, P, slope);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16raw>), dim3(ED, B), dim3(256), 0, ST, (const bf16raw*)fea, saved, gamma, beta, dgap, part2,
                           (bf16raw*)dfea, dgamma, dbeta, B, ED, P, slope);
    } else {
        hipLaunchKernelGGL((bn_bwd_stats_kernel<float>), dim3(ED, B), dim3(256), 0, ST, (const float*)fea, saved, gamma, beta, dgap, part2, B, ED, P, slope);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(ED, B), dim3(256), 0, ST, (const float*)fea, saved, gamma, beta, dgap, part2, (float*)dfea,
                           dgamma, dbeta, B, ED, P, slope);
    }
    FW_LAUNCH_RET();
}

extern "C" int fw_moco_logits(const float* q, const float* k, const float* queue, float* logits, float* khat, int L, int B, int ED, int K,
                              float invT, void* stream) {
    FW_CHECK_ARG(q && k && queue && logits && khat && L > 0 && B > 0 && ED > 0 && ED <= 1024 && K > 0);
    hipLaunchKernelGGL(moco_logits_kernel, dim3(L, B), dim3(256), (size_t)ED * 4, ST, q, k, queue, logits, khat, B, ED, K, invT);
    FW_LAUNCH_RET();
}
extern "C" int fw_moco_logits_bwd(const float* q, const float* khat, const float* queue, const float* dlogits, float* dq, int L, int B, int ED,
                                  int K, float invT, void* stream) {
    FW_CHECK_ARG(q && khat && queue && dlogits && dq && ED <= 1024);
    hipLaunchKernelGGL(moco_logits_bwd_kernel, dim3(L, B), dim3(256), 0, ST, q, khat, queue, dlogits, dq, B, ED, K, invT);
    FW_LAUNCH_RET();
}
extern "C" int fw_moco_enqueue(float* queue, const float* khat, long long* ptr, int L, int B, int ED, int K, void* stream) {
    FW_CHECK_ARG(queue && khat && ptr && K % B == 0);
    hipLaunchKernelGGL(moco_enqueue_kernel, dim3(fw_cdiv((long)L * B * ED, 256)), dim3(256), 0, ST, queue, khat, ptr, L, B, ED, K);
    hipLaunchKernelGGL(moco_ptr_kernel, dim3(1), dim3(1), 0, ST, ptr, B, K);
    FW_LAUNCH_RET();
}

// inter: f32 [nb1*B][NT][C] (bands 1.. of the encoder output, contiguous).  stats: f32 [nb1*B][NT][2].
extern "C" int fw_lfs_xbar(const float* inter, float* xbar, float* stats, int nb1, int B, int NT, int C, float eps, void* stream) {
    FW_CHECK_ARG(inter && xbar && stats && NT <= LFS_MAXT && NT > 0);
    hipLaunchKernelGGL(lfs_xbar_kernel, dim3(nb1 * B), dim3(256), 0, ST, inter, xbar, stats, NT, C, eps);
    FW_LAUNCH_RET();
}
extern "C" int fw_lfs_xbar_bwd(const float* inter, const float* stats, const float* dxbar, float* dinter, int nb1, int B, int NT, int C, void* stream) {
    FW_CHECK_ARG(inter && stats && dxbar && dinter);
    hipLaunchKernelGGL(lfs_xbar_bwd_kernel, dim3(nb1 * B), dim3(256), 0, ST, inter, stats, dxbar, dinter, NT, C);
    FW_LAUNCH_RET();
}
extern "C" int fw_lfs_lambda(const float* xbar, const unsigned long long* ptab, const int* heads, const long long* coef_off, float* coef,
                             float* save, int nblk, int B, int C, int nb1, void* stream) {
    FW_CHECK_ARG(xbar && ptab && heads && coef_off && coef && save && nblk > 0 && (nb1 == 1 || nb1 == 2));
    hipLaunchKernelGGL(lfs_lambda_kernel, dim3(nblk, B), dim3(64), 0, ST, xbar, ptab, heads, coef_off, coef, save, B, C, nb1);
    FW_LAUNCH_RET();
}
extern "C" int fw_lfs_lambda_bwd(const float* xbar, const unsigned long long* ptab, const unsigned long long* gtab, const int* heads,
                                 const long long* coef_off, const float* dcoef, const float* save, float* dxbar, int nblk, int B, int C,
                                 int nb1, void* stream) {
    FW_CHECK_ARG(xbar && ptab && gtab && heads && coef_off && dcoef && save && dxbar && (nb1 == 1 || nb1 == 2));
    hipLaunchKernelGGL(lfs_lambda_bwd_kernel, dim3(nblk, B), dim3(64), 0, ST, xbar, ptab, gtab, heads, coef_off, dcoef, save, dxbar, B, C, nb1);
    FW_LAUNCH_RET();
}
