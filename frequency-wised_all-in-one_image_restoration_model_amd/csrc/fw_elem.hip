// HBM-bound helpers of the AirNet hot path: LeFF depthwise conv (K4), im2col / col2im / pixel shuffle
// around the k4s2 and k2s2 convolutions (K7), the two 3-channel 3x3 projections, casts / copies,
// weight re-layouts, L1 / cross-entropy losses (train.py:88-92), fused Adam (train.py:63,96) and the MoCo
// EMA (net/utils/moco.py:44-50).  All tensors are token-major ("channels last"): row = (b, y, x).
// Everything here is bandwidth work: 16-byte accesses along channels, grid-stride loops, block-level
// partial sums before atomics.
#include "fw_common.h"

namespace {

constexpr int TPB = 256;
FW_DEV long gtid() { return (long)blockIdx.x * TPB + threadIdx.x; }
FW_DEV long gstride() { return (long)gridDim.x * TPB; }
static inline int grid_for(long n, int cap = 8192) { long g = (n + TPB - 1) / TPB; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

template <typename T> FW_DEV void ldvec(const T* p, float* f) {       // E16 elements
    unpack16<T>(*reinterpret_cast<const uint4*>(p), f);
}
template <typename T> FW_DEV void stvec(T* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack16<T>(f); }

// ------------------------------------------------------------------------------------------------
// cast / copy / axpy
// ------------------------------------------------------------------------------------------------
// dst[T][r][c] = src[f32][r][c] * (rowscale ? rowscale[r / rps] : 1)
template <typename T>
__global__ void cast_rows_kernel(const float* __restrict__ src, long lds_, T* __restrict__ dst, long ldd, long rows, int cols,
                                 const float* __restrict__ rowscale, int rps) {
    const int c4n = cols >> 2;
    for (long i = gtid(); i < rows * c4n; i += gstride()) {
        const long r = i / c4n; const int c = (int)(i % c4n) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds_ + c);
        const float s = rowscale ? rowscale[r / rps] : 1.f;
        T* d = dst + r * ldd + c;
        if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(d) = v * s;
        else *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf2(v[0] * s, v[1] * s), pack_bf2(v[2] * s, v[3] * s));
    }
}
// dst[f32][r][c] (=|+=) src[f32][r][c]
__global__ void copy_rows_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, long rows, int cols, int acc) {
    const int c4n = cols >> 2;
    for (long i = gtid(); i < rows * c4n; i += gstride()) {
        const long r = i / c4n; const int c = (int)(i % c4n) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds_ + c);
        if (acc) v += *reinterpret_cast<const f32x4*>(dst + r * ldd + c);
        *reinterpret_cast<f32x4*>(dst + r * ldd + c) = v;
    }
}
// dst[T] += src[T]   (rows x cols windows; used to fold the second key-gradient slot)
template <typename T>
__global__ void add_rows_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd, long rows, int cols) {
    const int c4n = cols >> 2;
    for (long i = gtid(); i < rows * c4n; i += gstride()) {
        const long r = i / c4n; const int c = (int)(i % c4n) * 4;
        for (int e = 0; e < 4; ++e) {
            T* d = dst + r * ldd + c + e;
            TT<T>::st(d, TT<T>::ld(d) + TT<T>::ld(src + r * lds_ + c + e));
        }
    }
}
template <typename T>
__global__ void cast_flat_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = gtid(); i < n; i += gstride()) TT<T>::st(dst + i, src[i]);
}
// generic 3-index permutation with conversion: out[o(a,b,c)] (=|+=) in[a][b][c]; so = output strides of (a,b,c)
template <typename TI, typename TO>
__global__ void permute3_kernel(const TI* __restrict__ in, TO* __restrict__ out, int d0, int d1, int d2, long s0, long s1, long s2, int acc) {
    const long n = (long)d0 * d1 * d2;
    for (long i = gtid(); i < n; i += gstride()) {
        const int c = (int)(i % d2); const long t = i / d2; const int b = (int)(t % d1); const int a = (int)(t / d1);
        const long o = a * s0 + b * s1 + c * s2;
        float v = TT<TI>::ld(in + i);
        if (acc) v += TT<TO>::ld(out + o);
        TT<TO>::st(out + o, v);
    }
}

// ------------------------------------------------------------------------------------------------
// LeFF depthwise 3x3 (net/utils/leff.py:104-111).  Pre- and post-activation tensors are both kept
// (h = pre, g = GELU(h)) so no erf is ever recomputed per tap:
//   fwd : h2 = dwconv3x3(g1) + bias ;  g2 = GELU(h2)
//   bwd : dh1 = GELU'(h1) * convT(dh2, w) ;  dw[c][tap] += sum_t g1[t+tap] dh2[t] ;  dbias[c] += sum_t dh2[t]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void dwconv_fwd_kernel(const T* __restrict__ g1, long ld1, const float* __restrict__ w, const float* __restrict__ bias,
                                  T* __restrict__ h2, T* __restrict__ g2, long ld2, int B, int H, int W, int C) {
    constexpr int E = TT<T>::E16;
    const int nv = C / E;
    const long total = (long)B * H * W * nv;
    for (long i = gtid(); i < total; i += gstride()) {
        const int v = (int)(i % nv); const long tok = i / nv;
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        const int c0 = v * E;
        float acc[E], wr[E][9];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            acc[e] = bias[c0 + e];
#pragma unroll
            for (int t = 0; t < 9; ++t) wr[e][t] = w[(c0 + e) * 9 + t];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                if (xx < 0 || xx >= W) continue;
                float f[E];
                ldvec<T>(g1 + ((b * H + yy) * W + xx) * ld1 + c0, f);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += f[e] * wr[e][ky * 3 + kx];
            }
        }
        stvec<T>(h2 + tok * ld2 + c0, acc);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = gelu_f(acc[e]);
        stvec<T>(g2 + tok * ld2 + c0, acc);
    }
}

// block = 32 token stripes x 8 channel vectors; a thread walks STRIPE consecutive tokens of one vector with its
// weight-gradient partials in registers; partials are reduced over the block's 32 stripes (shuffles + LDS) before ONE
// atomic per (channel, tap) and block -- every block of a layer hits the same few hundred words otherwise.
template <typename T, int STRIPE>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ dh2, long ldg, const T* __restrict__ g1, const T* __restrict__ h1,
                                                         long ld1, const float* __restrict__ w, T* __restrict__ dh1, long ldo,
                                                         float* __restrict__ dw, float* __restrict__ dbias, int B, int H, int W, int C) {
    constexpr int E = TT<T>::E16;
    __shared__ float red[8 * E * 10];
    const int nv = C / E;
    const int nvg = (nv + 7) / 8;
    const long ntok = (long)B * H * W;
    const int vl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int v = (blockIdx.x % nvg) * 8 + vl;
    const long st = (long)(blockIdx.x / nvg) * 32 + sl;
    const bool live = v < nv;
    const int c0 = (live ? v : 0) * E;
    for (int i = threadIdx.x; i < 8 * E * 10; i += 256) red[i] = 0.f;
    float wr[E][9], gw[E][10];
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int t = 0; t < 9; ++t) { wr[e][t] = w[(c0 + e) * 9 + t]; gw[e][t] = 0.f; }
        gw[e][9] = 0.f;
    }
    if (live)
        for (long tok = st * STRIPE; tok < ntok && tok < (st + 1) * STRIPE; ++tok) {
            const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
            float g0[E], hc[E], din[E];
            ldvec<T>(dh2 + tok * ldg + c0, g0);
            ldvec<T>(h1 + tok * ld1 + c0, hc);
#pragma unroll
            for (int e = 0; e < E; ++e) { gw[e][9] += g0[e]; din[e] = 0.f; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = x + kx - 1;          // weight gradient: output t x input t + (ky-1, kx-1)
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        float f[E];
                        ldvec<T>(g1 + ((b * H + yy) * W + xx) * ld1 + c0, f);
#pragma unroll
                        for (int e = 0; e < E; ++e) gw[e][ky * 3 + kx] += f[e] * g0[e];
                    }
                    const int yo = y - ky + 1, xo = x - kx + 1;          // data gradient: input t <- w[tap] * dh2[t - (ky-1, kx-1)]
                    if (yo >= 0 && yo < H && xo >= 0 && xo < W) {
                        float f[E];
                        ldvec<T>(dh2 + ((b * H + yo) * W + xo) * ldg + c0, f);
#pragma unroll
                        for (int e = 0; e < E; ++e) din[e] += wr[e][ky * 3 + kx] * f[e];
                    }
                }
#pragma unroll
            for (int e = 0; e < E; ++e) din[e] *= gelu_grad_f(hc[e]);
            stvec<T>(dh1 + tok * ldo + c0, din);
        }
    __syncthreads();
    // reduce over the 8 stripes of this wave (lanes with equal vl), then over the 4 waves through LDS
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float s = gw[e][t];
            s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            if ((threadIdx.x & 63) < 8) atomicAdd(&red[(vl * E + e) * 10 + t], s);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * E * 10; i += 256) {
        const int t = i % 10, ce = i / 10;
        const int c = (blockIdx.x % nvg) * 8 * E + ce;
        if (c < C) {
            if (t < 9) atomicAdd(dw + c * 9 + t, red[i]);
            else atomicAdd(dbias + c, red[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k4 s2 p1 convolution as GEMM: im2col (f32 stream -> T, K order (ky,kx,ci)) and its adjoint
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col4_kernel(const float* __restrict__ x, long ldx, T* __restrict__ col, int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4n = C >> 2;
    const long total = (long)B * Ho * Wo * 16 * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; long t = i / c4n;
        const int tap = (int)(t % 16); t /= 16;
        const int ox = (int)(t % Wo); const int oy = (int)((t / Wo) % Ho); const long b = t / ((long)Wo * Ho);
        const int iy = 2 * oy - 1 + (tap >> 2), ix = 2 * ox - 1 + (tap & 3);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(x + ((b * H + iy) * W + ix) * ldx + c);
        T* d = col + (t * 16 + tap) * (long)C + c;
        if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(d) = v;
        else *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    }
}
// dx[f32][tin][c] = (dres?) + sum over the <= 4 (output pixel, tap) pairs that read input pixel tin
template <typename T>
__global__ void col2im4_kernel(const T* __restrict__ dcol, float* __restrict__ dx, long lddx, const float* __restrict__ dres, long ldr,
                               int B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, c4n = C >> 2;
    const long total = (long)B * H * W * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; const long tok = i / c4n;
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (dres) acc = *reinterpret_cast<const f32x4*>(dres + tok * ldr + c);
        for (int ky = (y + 1) & 1; ky < 4; ky += 2) {
            const int oy = (y + 1 - ky) / 2;
            if (y + 1 - ky < 0 || oy >= Ho) continue;
            for (int kx = (x + 1) & 1; kx < 4; kx += 2) {
                const int ox = (x + 1 - kx) / 2;
                if (x + 1 - kx < 0 || ox >= Wo) continue;
                const T* s = dcol + (((b * Ho + oy) * Wo + ox) * 16 + ky * 4 + kx) * (long)C + c;
                for (int e = 0; e < 4; ++e) acc[e] += TT<T>::ld(s + e);
            }
        }
        *reinterpret_cast<f32x4*>(dx + tok * lddx + c) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// k2 s2 transposed convolution = Linear(Cin -> 4*Cout) + depth-to-space
// ------------------------------------------------------------------------------------------------
// out[f32][(b,2y+i,2x+j)][co] = g[T][(b,y,x)][(i*2+j)*Cout + co] + bias[co]
template <typename T>
__global__ void pixel_shuffle_kernel(const T* __restrict__ g, const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                     int B, int H, int W, int Cout) {
    const int c4n = Cout >> 2;
    const long total = (long)B * H * W * 4 * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; long t = i / c4n;
        const int q = (int)(t % 4); t /= 4;
        const int x = (int)(t % W); const int y = (int)((t / W) % H); const long b = t / ((long)W * H);
        const T* s = g + t * 4L * Cout + q * Cout + c;
        float* d = out + ((b * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * ldo + c;
        f32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = TT<T>::ld(s + e) + bias[c + e];
        *reinterpret_cast<f32x4*>(d) = v;
    }
}
// dg[T][(b,y,x)][(i*2+j)*Cout + co] = dout[f32][(b,2y+i,2x+j)][co];  dbias[co] += sum
template <typename T>
__global__ void pixel_unshuffle_kernel(const float* __restrict__ dout, long ldo, T* __restrict__ dg, int B, int H, int W, int Cout) {
    const int c4n = Cout >> 2;
    const long total = (long)B * H * W * 4 * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; long t = i / c4n;
        const int q = (int)(t % 4); t /= 4;
        const int x = (int)(t % W); const int y = (int)((t / W) % H); const long b = t / ((long)W * H);
        const float* s = dout + ((b * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * ldo + c;
        T* d = dg + t * 4L * Cout + q * Cout + c;
        for (int e = 0; e < 4; ++e) TT<T>::st(d + e, s[e]);
    }
}
// column sums of an f32 matrix: out[c] += sum_r x[r][c]   (bias gradients)
__global__ void colsum_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, long rows, int cols, int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (int c = threadIdx.x; c < cols; c += TPB) {
        float s = 0.f;
        for (long r = r0; r < rows && r < r0 + rows_per_block; ++r) s += x[r * ldx + c];
        atomicAdd(out + c, s);
    }
}
template <typename T>
__global__ void colsum_t_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, long rows, int cols, int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    for (int c = threadIdx.x; c < cols; c += TPB) {
        float s = 0.f;
        for (long r = r0; r < rows && r < r0 + rows_per_block; ++r) s += TT<T>::ld(x + r * ldx + c);
        atomicAdd(out + c, s);
    }
}

// ------------------------------------------------------------------------------------------------
// InputProj: 3x3 conv 3 -> C + LeakyReLU(0.01) on an NCHW f32 image  (decoder_Uformer.py:453-472)
// ------------------------------------------------------------------------------------------------
__global__ void inproj_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ out, long ldo, int B, int H, int W, int C, float slope) {
    const int c4n = C >> 2;
    const long total = (long)B * H * W * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; const long tok = i / c4n;
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
        for (int ci = 0; ci < 3; ++ci)
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if (xx < 0 || xx >= W) continue;
                    const float p = img[((b * 3 + ci) * H + yy) * W + xx];
                    for (int e = 0; e < 4; ++e) acc[e] += p * w[((c + e) * 3 + ci) * 9 + ky * 3 + kx];
                }
            }
        for (int e = 0; e < 4; ++e) acc[e] = lrelu_f(acc[e], slope);
        *reinterpret_cast<f32x4*>(out + tok * ldo + c) = acc;
    }
}
// dw[c][ci][ky][kx] += sum_t dy'[t][c] img[..];  db[c] += sum_t dy'[t][c];   dy' = dy * lrelu'(out)
template <int STRIPE>
__global__ __launch_bounds__(256) void inproj_bwd_kernel(const float* __restrict__ img, const float* __restrict__ out, long ldo, const float* __restrict__ dy, long ldy,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int H, int W, int C, float slope) {
    extern __shared__ float red[];                       // [C][28]: block partials before ONE atomic per word
    for (int i = threadIdx.x; i < C * 28; i += 256) red[i] = 0.f;
    __syncthreads();
    const long ntok = (long)B * H * W;
    const long nstripes = (ntok + STRIPE - 1) / STRIPE;
    for (long i = gtid(); i < nstripes * C; i += gstride()) {
        const int c = (int)(i % C); const long st = i / C;
        float g[27], gb = 0.f;
        for (int t = 0; t < 27; ++t) g[t] = 0.f;
        for (long tok = st * STRIPE; tok < ntok && tok < (st + 1) * STRIPE; ++tok) {
            const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
            float d = dy[tok * ldy + c];
            if (out[tok * ldo + c] <= 0.f) d *= slope;
            gb += d;
            for (int ci = 0; ci < 3; ++ci)
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= W) continue;
                        g[ci * 9 + ky * 3 + kx] += d * img[((b * 3 + ci) * H + yy) * W + xx];
                    }
                }
        }
        atomicAdd(&red[c * 28 + 27], gb);
        for (int t = 0; t < 27; ++t) atomicAdd(&red[c * 28 + t], g[t]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 28; i += 256) {
        const int c = i / 28, t = i % 28;
        if (t < 27) atomicAdd(dw + c * 27 + t, red[i]); else atomicAdd(db + c, red[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// OutputProj: 3x3 conv C -> 3 on tokens, + global residual, NCHW f32 out  (decoder_Uformer.py:476-499,1171)
// ------------------------------------------------------------------------------------------------
__global__ void outproj_fwd_kernel(const float* __restrict__ fea, long ldf, const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ img, float* __restrict__ out, int B, int H, int W, int C) {
    const long total = (long)B * H * W;
    for (long tok = gtid(); tok < total; tok += gstride()) {
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        float a0 = bias[0], a1 = bias[1], a2 = bias[2];
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                if (xx < 0 || xx >= W) continue;
                const float* f = fea + ((b * H + yy) * W + xx) * ldf;
                const int tap = ky * 3 + kx;
                for (int c = 0; c < C; c += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(f + c);
                    for (int e = 0; e < 4; ++e) {
                        a0 += v[e] * w[(0 * C + c + e) * 9 + tap];
                        a1 += v[e] * w[(1 * C + c + e) * 9 + tap];
                        a2 += v[e] * w[(2 * C + c + e) * 9 + tap];
                    }
                }
            }
        }
        const long p = (b * 3 * H + y) * W + x;
        out[p] = a0 + (img ? img[p] : 0.f);
        out[p + (long)H * W] = a1 + (img ? img[p + (long)H * W] : 0.f);
        out[p + 2L * H * W] = a2 + (img ? img[p + 2L * H * W] : 0.f);
    }
}
// dfea[tok][c] = sum_{co,tap} dout[co][tok - tap] w[co][c][tap]
__global__ void outproj_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w, float* __restrict__ dfea, long ldf,
                                        int B, int H, int W, int C) {
    const int c4n = C >> 2;
    const long total = (long)B * H * W * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; const long tok = i / c4n;
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < 3; ++ky) {
            const int yo = y - ky + 1;
            if (yo < 0 || yo >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xo = x - kx + 1;
                if (xo < 0 || xo >= W) continue;
                for (int co = 0; co < 3; ++co) {
                    const float d = dout[((b * 3 + co) * H + yo) * W + xo];
                    for (int e = 0; e < 4; ++e) acc[e] += d * w[(co * C + c + e) * 9 + ky * 3 + kx];
                }
            }
        }
        *reinterpret_cast<f32x4*>(dfea + tok * ldf + c) = acc;
    }
}
template <int STRIPE>
__global__ __launch_bounds__(256) void outproj_bwd_w_kernel(const float* __restrict__ dout, const float* __restrict__ fea, long ldf, float* __restrict__ dw,
                                                            float* __restrict__ db, int B, int H, int W, int C) {
    extern __shared__ float red[];                       // [C][27] + [3]
    for (int i = threadIdx.x; i < C * 27 + 3; i += 256) red[i] = 0.f;
    __syncthreads();
    const long ntok = (long)B * H * W;
    const long nstripes = (ntok + STRIPE - 1) / STRIPE;
    for (long i = gtid(); i < nstripes * C; i += gstride()) {
        const int c = (int)(i % C); const long st = i / C;
        float g[27], gb[3] = {0.f, 0.f, 0.f};
        for (int t = 0; t < 27; ++t) g[t] = 0.f;
        for (long tok = st * STRIPE; tok < ntok && tok < (st + 1) * STRIPE; ++tok) {
            const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
            float d[3];
            for (int co = 0; co < 3; ++co) { d[co] = dout[((b * 3 + co) * H + y) * W + x]; gb[co] += d[co]; }
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if (xx < 0 || xx >= W) continue;
                    const float f = fea[((b * H + yy) * W + xx) * ldf + c];
                    for (int co = 0; co < 3; ++co) g[co * 9 + ky * 3 + kx] += d[co] * f;
                }
            }
        }
        if (c == 0) for (int co = 0; co < 3; ++co) atomicAdd(&red[C * 27 + co], gb[co]);
        for (int t = 0; t < 27; ++t) atomicAdd(&red[c * 27 + t], g[t]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 27 + 3; i += 256) {
        if (i >= C * 27) { atomicAdd(db + (i - C * 27), red[i]); continue; }
        const int c = i / 27, t = i % 27;                 // t = co*9 + tap
        atomicAdd(dw + ((t / 9) * C + c) * 9 + t % 9, red[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// losses
// ------------------------------------------------------------------------------------------------
// loss += mean |a - b| ; dA = sign(a - b) * gscale / n
__global__ void l1_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da, long n, float gscale,
                               float* __restrict__ loss) {
    float s = 0.f;
    for (long i = gtid(); i < n; i += gstride()) {
        const float d = a[i] - b[i];
        s += fabsf(d);
        if (da) da[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * gscale / n;
    }
    s = wave_sum(s);
    if (lane_id() == 0) atomicAdd(loss, s / n);
}
// cross entropy with label 0 over rows of logits[R][N]: loss += mean_r (lse - logit0); dlogits = (softmax - onehot0) * gscale / R
__global__ void ce0_loss_kernel(const float* __restrict__ logits, float* __restrict__ dlogits, int R, int N, float gscale, float* __restrict__ loss) {
    const int r = blockIdx.x;
    const float* x = logits + (long)r * N;
    float mx = -3.0e38f;
    for (int j = threadIdx.x; j < N; j += 64) mx = fmaxf(mx, x[j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 64) s += __expf(x[j] - mx);
    s = wave_sum(s);
    const float lse = mx + __logf(s);
    if (dlogits)
        for (int j = threadIdx.x; j < N; j += 64) dlogits[(long)r * N + j] = (__expf(x[j] - lse) - (j == 0 ? 1.f : 0.f)) * gscale / R;
    if (threadIdx.x == 0) atomicAdd(loss, (lse - x[0]) / R);
}

// ------------------------------------------------------------------------------------------------
// optimizer: Adam (torch.optim.Adam defaults, train.py:63) + low-precision shadow; MoCo EMA
// ------------------------------------------------------------------------------------------------
// hyper (device, f32[4]): lr, beta1^t, beta2^t, -- ; advanced by adam_tick_kernel so that a captured
// HIP graph replays the right bias correction and learning rate every step.
__global__ void adam_tick_kernel(float* __restrict__ hyper, float b1, float b2) { hyper[1] *= b1; hyper[2] *= b2; }
template <typename T>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            T* __restrict__ shadow, long n, const float* __restrict__ hyper, float b1, float b2, float eps) {
    const float lr = hyper[0], bc1 = 1.f - hyper[1], bc2_sqrt = sqrtf(1.f - hyper[2]);
    for (long i = gtid(); i < n; i += gstride()) {
        const float gi = g[i];
        const float mi = m[i] * b1 + gi * (1.f - b1);
        const float vi = v[i] * b2 + gi * gi * (1.f - b2);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pi = p[i] - (lr / bc1) * (mi / denom);
        p[i] = pi;
        if (shadow) TT<T>::st(shadow + i, pi);
    }
}
template <typename T>
__global__ void ema_kernel(float* __restrict__ pk, const float* __restrict__ pq, T* __restrict__ shadow, long n, float mom) {
    for (long i = gtid(); i < n; i += gstride()) {
        const float x = pk[i] * mom + pq[i] * (1.f - mom);
        pk[i] = x;
        if (shadow) TT<T>::st(shadow + i, x);
    }
}
// y[T] = lrelu(x[f32]);  dx[f32] = dy[T] * (y > 0 ? 1 : slope)      (encoder head MLPs, encoder_Uformer.py:953-957)
template <typename T>
__global__ void lrelu_fwd_kernel(const float* __restrict__ x, T* __restrict__ y, long n, float slope) {
    for (long i = gtid(); i < n; i += gstride()) TT<T>::st(y + i, lrelu_f(x[i], slope));
}
template <typename T>
__global__ void lrelu_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long n, float slope) {
    for (long i = gtid(); i < n; i += gstride()) dx[i] = TT<T>::ld(dy + i) * (x[i] > 0.f ? 1.f : slope);
}
// dst[i] (+)= sum_z slab[z*zstride + i]: the reduction step of a split-K GEMM (replaces thousands of same-address atomics)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int nz, long n, long zstride, float* __restrict__ dst, int accumulate, int zper) {
    const long i4 = gtid();
    if (i4 * 4 >= n) return;
    const int z0 = blockIdx.y * zper, z1 = min(nz, z0 + zper);
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i4 * 4 + 3 < n) {
        for (int z = z0; z < z1; ++z) s += *reinterpret_cast<const f32x4*>(slab + (long)z * zstride + i4 * 4);
    } else {
        for (int z = z0; z < z1; ++z)
            for (int e = 0; e < 4; ++e) if (i4 * 4 + e < n) s[e] += slab[(long)z * zstride + i4 * 4 + e];
    }
    for (int e = 0; e < 4; ++e) {
        if (i4 * 4 + e >= n) break;
        if (gridDim.y > 1) atomicAdd(dst + i4 * 4 + e, s[e]);        // <= nz/zper adders per address
        else if (accumulate) dst[i4 * 4 + e] += s[e];
        else dst[i4 * 4 + e] = s[e];
    }
}
__global__ void fill_kernel(float* __restrict__ p, long n, float v) {
    for (long i = gtid(); i < n; i += gstride()) p[i] = v;
}

}  // namespace

#define ST ((hipStream_t)stream)
#define LAUNCH(kern, n, ...)                                                           \
    do {                                                                               \
        hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(TPB), 0, ST, __VA_ARGS__);    \
        FW_LAUNCH_RET();                                                               \
    } while (0)
#define BYT(dtype, expr_bf, expr_f) return (dtype) == FW_DT_BF16 ? (expr_bf) : (expr_f)

extern "C" int fw_cast_rows(int dtype, const float* src, long lds_, void* dst, long ldd, long rows, int cols,
                            const float* rowscale, int rows_per_scale, void* stream) {
    FW_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0);
    if (dtype == FW_DT_BF16) LAUNCH((cast_rows_kernel<bf16raw>), rows * (cols / 4), src, lds_, (bf16raw*)dst, ldd, rows, cols, rowscale, rows_per_scale);
    LAUNCH((cast_rows_kernel<float>), rows * (cols / 4), src, lds_, (float*)dst, ldd, rows, cols, rowscale, rows_per_scale);
}
extern "C" int fw_copy_rows(const float* src, long lds_, float* dst, long ldd, long rows, int cols, int accumulate, void* stream) {
    FW_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0);
    LAUNCH(copy_rows_kernel, rows * (cols / 4), src, lds_, dst, ldd, rows, cols, accumulate);
}
extern "C" int fw_add_rows(int dtype, const void* src, long lds_, void* dst, long ldd, long rows, int cols, void* stream) {
    FW_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols % 4 == 0);
    if (dtype == FW_DT_BF16) LAUNCH((add_rows_kernel<bf16raw>), rows * (cols / 4), (const bf16raw*)src, lds_, (bf16raw*)dst, ldd, rows, cols);
    LAUNCH((add_rows_kernel<float>), rows * (cols / 4), (const float*)src, lds_, (float*)dst, ldd, rows, cols);
}
extern "C" int fw_cast_flat(int dtype, const float* src, void* dst, long n, void* stream) {
    FW_CHECK_ARG(src && dst && n > 0);
    if (dtype == FW_DT_BF16) LAUNCH((cast_flat_kernel<bf16raw>), n, src, (bf16raw*)dst, n);
    LAUNCH((cast_flat_kernel<float>), n, src, (float*)dst, n);
}
// in_dtype/out_dtype: 0 f32, 1 bf16.  out[a*s0 + b*s1 + c*s2] (=|+=) in[a][b][c]
extern "C" int fw_permute3(int in_dtype, int out_dtype, const void* in, void* out, int d0, int d1, int d2, long s0, long s1,
                           long s2, int accumulate, void* stream) {
    FW_CHECK_ARG(in && out && d0 > 0 && d1 > 0 && d2 > 0);
    const long n = (long)d0 * d1 * d2;
    if (in_dtype == 0 && out_dtype == 0) LAUNCH((permute3_kernel<float, float>), n, (const float*)in, (float*)out, d0, d1, d2, s0, s1, s2, accumulate);
    if (in_dtype == 0 && out_dtype == 1) LAUNCH((permute3_kernel<float, bf16raw>), n, (const float*)in, (bf16raw*)out, d0, d1, d2, s0, s1, s2, accumulate);
    if (in_dtype == 1 && out_dtype == 0) LAUNCH((permute3_kernel<bf16raw, float>), n, (const bf16raw*)in, (float*)out, d0, d1, d2, s0, s1, s2, accumulate);
    return -1;
}
extern "C" int fw_dwconv_fwd(int dtype, const void* g1, long ld1, const float* w, const float* bias, void* h2, void* g2, long ld2,
                             int B, int H, int W, int C, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(g1 && w && bias && h2 && g2 && C % e == 0 && ld1 % e == 0 && ld2 % e == 0);
    const long n = (long)B * H * W * (C / e);
    if (dtype == FW_DT_BF16) LAUNCH((dwconv_fwd_kernel<bf16raw>), n, (const bf16raw*)g1, ld1, w, bias, (bf16raw*)h2, (bf16raw*)g2, ld2, B, H, W, C);
    LAUNCH((dwconv_fwd_kernel<float>), n, (const float*)g1, ld1, w, bias, (float*)h2, (float*)g2, ld2, B, H, W, C);
}
extern "C" int fw_dwconv_bwd(int dtype, const void* dh2, long ldg, const void* g1, const void* h1, long ld1, const float* w, void* dh1,
                             long ldo, float* dw, float* dbias, int B, int H, int W, int C, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(dh2 && g1 && h1 && w && dh1 && dw && dbias && C % e == 0 && ld1 % e == 0 && ldg % e == 0 && ldo % e == 0);
    constexpr int STRIPE = 32;
    const int nvg = (C / e + 7) / 8;
    const long nsg = (((long)B * H * W + STRIPE - 1) / STRIPE + 31) / 32;
    const dim3 grid((unsigned)(nsg * nvg));
    if (dtype == FW_DT_BF16)
        hipLaunchKernelGGL((dwconv_bwd_kernel<bf16raw, STRIPE>), grid, dim3(256), 0, ST, (const bf16raw*)dh2, ldg, (const bf16raw*)g1,
                           (const bf16raw*)h1, ld1, w, (bf16raw*)dh1, ldo, dw, dbias, B, H, W, C);
    else
        hipLaunchKernelGGL((dwconv_bwd_kernel<float, STRIPE>), grid, dim3(256), 0, ST, (const float*)dh2, ldg, (const float*)g1,
                           (const float*)h1, ld1, w, (float*)dh1, ldo, dw, dbias, B, H, W, C);
    FW_LAUNCH_RET();
}
extern "C" int fw_im2col4(int dtype, const float* x, long ldx, void* col, int B, int H, int W, int C, void* stream) {
    FW_CHECK_ARG(x && col && C % 4 == 0 && ldx % 4 == 0 && H % 2 == 0 && W % 2 == 0);
    const long n = (long)B * (H / 2) * (W / 2) * 16 * (C / 4);
    if (dtype == FW_DT_BF16) LAUNCH((im2col4_kernel<bf16raw>), n, x, ldx, (bf16raw*)col, B, H, W, C);
    LAUNCH((im2col4_kernel<float>), n, x, ldx, (float*)col, B, H, W, C);
}
extern "C" int fw_col2im4(int dtype, const void* dcol, float* dx, long lddx, const float* dres, long ldr, int B, int H, int W, int C,
                          void* stream) {
    FW_CHECK_ARG(dcol && dx && C % 4 == 0 && lddx % 4 == 0 && H % 2 == 0 && W % 2 == 0 && (!dres || ldr % 4 == 0));
    const long n = (long)B * H * W * (C / 4);
    if (dtype == FW_DT_BF16) LAUNCH((col2im4_kernel<bf16raw>), n, (const bf16raw*)dcol, dx, lddx, dres, ldr, B, H, W, C);
    LAUNCH((col2im4_kernel<float>), n, (const float*)dcol, dx, lddx, dres, ldr, B, H, W, C);
}
extern "C" int fw_pixel_shuffle(int dtype, const void* g, const float* bias, float* out, long ldo, int B, int H, int W, int Cout,
                                void* stream) {
    FW_CHECK_ARG(g && bias && out && Cout % 4 == 0 && ldo % 4 == 0);
    const long n = (long)B * H * W * Cout;
    if (dtype == FW_DT_BF16) LAUNCH((pixel_shuffle_kernel<bf16raw>), n, (const bf16raw*)g, bias, out, ldo, B, H, W, Cout);
    LAUNCH((pixel_shuffle_kernel<float>), n, (const float*)g, bias, out, ldo, B, H, W, Cout);
}
extern "C" int fw_pixel_unshuffle(int dtype, const float* dout, long ldo, void* dg, int B, int H, int W, int Cout, void* stream) {
    FW_CHECK_ARG(dout && dg && Cout % 4 == 0 && ldo % 4 == 0);
    const long n = (long)B * H * W * Cout;
    if (dtype == FW_DT_BF16) LAUNCH((pixel_unshuffle_kernel<bf16raw>), n, dout, ldo, (bf16raw*)dg, B, H, W, Cout);
    LAUNCH((pixel_unshuffle_kernel<float>), n, dout, ldo, (float*)dg, B, H, W, Cout);
}
// out[c] += sum_r x[r][c];  x_dtype: 0 f32 / 1 bf16
extern "C" int fw_colsum(int x_dtype, const void* x, long ldx, float* out, long rows, int cols, void* stream) {
    FW_CHECK_ARG(x && out && rows > 0 && cols > 0);
    int rpb = (int)((rows + 1023) / 1024);
    if (rpb < 32) rpb = 32;
    const int grid = (int)((rows + rpb - 1) / rpb);
    if (x_dtype == 1) hipLaunchKernelGGL((colsum_t_kernel<bf16raw>), dim3(grid), dim3(TPB), 0, ST, (const bf16raw*)x, ldx, out, rows, cols, rpb);
    else hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(TPB), 0, ST, (const float*)x, ldx, out, rows, cols, rpb);
    FW_LAUNCH_RET();
}
extern "C" int fw_inproj_fwd(const float* img, const float* w, const float* bias, float* out, long ldo, int B, int H, int W, int C,
                             float slope, void* stream) {
    FW_CHECK_ARG(img && w && bias && out && C % 4 == 0 && ldo % 4 == 0);
    LAUNCH(inproj_fwd_kernel, (long)B * H * W * (C / 4), img, w, bias, out, ldo, B, H, W, C, slope);
}
extern "C" int fw_inproj_bwd(const float* img, const float* out, long ldo, const float* dy, long ldy, float* dw, float* db, int B,
                             int H, int W, int C, float slope, void* stream) {
    FW_CHECK_ARG(img && out && dy && dw && db);
    constexpr int STRIPE = 32;
    hipLaunchKernelGGL((inproj_bwd_kernel<STRIPE>), dim3(grid_for((((long)B * H * W + STRIPE - 1) / STRIPE) * C, 1024)), dim3(TPB), (size_t)C * 28 * 4, ST,
                       img, out, ldo, dy, ldy, dw, db, B, H, W, C, slope);
    FW_LAUNCH_RET();
}
extern "C" int fw_outproj_fwd(const float* fea, long ldf, const float* w, const float* bias, const float* img, float* out, int B,
                              int H, int W, int C, void* stream) {
    FW_CHECK_ARG(fea && w && bias && out && C % 4 == 0 && ldf % 4 == 0);
    LAUNCH(outproj_fwd_kernel, (long)B * H * W, fea, ldf, w, bias, img, out, B, H, W, C);
}
extern "C" int fw_outproj_bwd(const float* dout, const float* fea, long ldf, const float* w, float* dfea, long lddf, float* dw,
                              float* db, int B, int H, int W, int C, void* stream) {
    FW_CHECK_ARG(dout && fea && w && dfea && dw && db && C % 4 == 0 && ldf % 4 == 0 && lddf % 4 == 0);
    hipLaunchKernelGGL(outproj_bwd_data_kernel, dim3(grid_for((long)B * H * W * (C / 4))), dim3(TPB), 0, ST, dout, w, dfea, lddf, B, H, W, C);
    constexpr int STRIPE = 32;
    hipLaunchKernelGGL((outproj_bwd_w_kernel<STRIPE>), dim3(grid_for((((long)B * H * W + STRIPE - 1) / STRIPE) * C, 1024)), dim3(TPB),
                       (size_t)(C * 27 + 3) * 4, ST, dout, fea, ldf, dw, db, B, H, W, C);
    FW_LAUNCH_RET();
}
extern "C" int fw_l1_loss(const float* a, const float* b, float* da, long n, float gscale, float* loss, void* stream) {
    FW_CHECK_ARG(a && b && loss && n > 0);
    hipLaunchKernelGGL(l1_loss_kernel, dim3(grid_for(n, 1024)), dim3(TPB), 0, ST, a, b, da, n, gscale, loss);
    FW_LAUNCH_RET();
}
extern "C" int fw_ce0_loss(const float* logits, float* dlogits, int R, int N, float gscale, float* loss, void* stream) {
    FW_CHECK_ARG(logits && loss && R > 0 && N > 0);
    hipLaunchKernelGGL(ce0_loss_kernel, dim3(R), dim3(64), 0, ST, logits, dlogits, R, N, gscale, loss);
    FW_LAUNCH_RET();
}
// hyper: device f32[4] = {lr, beta1^t, beta2^t, 0}; initialise to {lr, 1, 1, 0}.  fw_adam_tick advances t by one.
extern "C" int fw_adam_tick(float* hyper, float b1, float b2, void* stream) {
    FW_CHECK_ARG(hyper);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, ST, hyper, b1, b2);
    FW_LAUNCH_RET();
}
extern "C" int fw_adam(int shadow_dtype, float* p, const float* g, float* m, float* v, void* shadow, long n, const float* hyper,
                       float b1, float b2, float eps, void* stream) {
    FW_CHECK_ARG(p && g && m && v && hyper && n > 0);
    if (shadow_dtype == FW_DT_BF16 && shadow) LAUNCH((adam_kernel<bf16raw>), n, p, g, m, v, (bf16raw*)shadow, n, hyper, b1, b2, eps);
    LAUNCH((adam_kernel<float>), n, p, g, m, v, (float*)nullptr, n, hyper, b1, b2, eps);
}
extern "C" int fw_ema(int shadow_dtype, float* pk, const float* pq, void* shadow, long n, float momentum, void* stream) {
    FW_CHECK_ARG(pk && pq && n > 0);
    if (shadow_dtype == FW_DT_BF16 && shadow) LAUNCH((ema_kernel<bf16raw>), n, pk, pq, (bf16raw*)shadow, n, momentum);
    LAUNCH((ema_kernel<float>), n, pk, pq, (float*)nullptr, n, momentum);
}
extern "C" int fw_lrelu_fwd(int dtype, const float* x, void* y, long n, float slope, void* stream) {
    FW_CHECK_ARG(x && y && n > 0);
    if (dtype == FW_DT_BF16) LAUNCH((lrelu_fwd_kernel<bf16raw>), n, x, (bf16raw*)y, n, slope);
    LAUNCH((lrelu_fwd_kernel<float>), n, x, (float*)y, n, slope);
}
extern "C" int fw_lrelu_bwd(int dtype, const void* dy, const float* x, float* dx, long n, float slope, void* stream) {
    FW_CHECK_ARG(dy && x && dx && n > 0);
    if (dtype == FW_DT_BF16) LAUNCH((lrelu_bwd_kernel<bf16raw>), n, (const bf16raw*)dy, x, dx, n, slope);
    LAUNCH((lrelu_bwd_kernel<float>), n, (const float*)dy, x, dx, n, slope);
}
// dst must be pre-initialised when accumulate != 0 or when nz > 64 (the z range is then split over blockIdx.y with atomics).
extern "C" int fw_slab_reduce(const float* slab, int nz, long n, long zstride, float* dst, int accumulate, void* stream) {
    FW_CHECK_ARG(slab && dst && nz > 0 && n > 0 && zstride % 4 == 0 && ((uintptr_t)slab & 15) == 0);
    FW_CHECK_ARG(accumulate || nz <= 64);
    const int zper = 64;
    dim3 grid((unsigned)((n / 4 + 1 + TPB - 1) / TPB), (unsigned)((nz + zper - 1) / zper));
    hipLaunchKernelGGL(slab_reduce_kernel, grid, dim3(TPB), 0, ST, slab, nz, n, zstride, dst, accumulate, zper);
    FW_LAUNCH_RET();
}
extern "C" int fw_fill(float* p, long n, float v, void* stream) {
    FW_CHECK_ARG(p && n > 0);
    LAUNCH(fill_kernel, n, p, n, v);
}
