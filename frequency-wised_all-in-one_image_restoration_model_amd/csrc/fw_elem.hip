// HBM-bound helpers of the AirNet hot path: LeFF depthwise conv (K4), im2col / col2im / pixel shuffle
// around the k4s2 and k2s2 convolutions (K7), the two 3-channel 3x3 projections, casts / copies,
// weight re-layouts, L1 / cross-entropy losses (train.py:88-92), fused Adam (train.py:63,96) and the MoCo
// EMA (net/utils/moco.py:44-50).  All tensors are token-major ("channels last"): row = (b, y, x).
// Everything here is bandwidth work: 16-byte accesses along channels, grid-stride loops, block-level
// partial sums before atomics.
#include "fw_common.h"
#include <stdlib.h>

namespace {

constexpr int TPB = 256;
FW_DEV long gtid() { return (long)blockIdx.x * TPB + threadIdx.x; }
FW_DEV long gstride() { return (long)gridDim.x * TPB; }
static inline int grid_for(long n, int cap = 8192) { long g = (n + TPB - 1) / TPB; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

template <typename T> FW_DEV void ldvec(const T* p, float* f) {       // E16 elements
    unpack16<T>(*reinterpret_cast<const uint4*>(p), f);
}
template <typename T> FW_DEV void stvec(T* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack16<T>(f); }
// 4-element vectors (16 B of f32 / 8 B of bf16): half the registers per thread of the 16-byte form -> twice the waves in flight
template <typename T> FW_DEV void ldvec4(const T* p, float* f);
template <> FW_SPEC void ldvec4<float>(const float* p, float* f) { unpack16<float>(*reinterpret_cast<const uint4*>(p), f); }
template <> FW_SPEC void ldvec4<bf16raw>(const bf16raw* p, float* f) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> FW_DEV void stvec4(T* p, const float* f);
template <> FW_SPEC void stvec4<float>(float* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack16<float>(f); }
template <> FW_SPEC void stvec4<bf16raw>(bf16raw* p, const float* f) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]));
}

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

// Many small f32 -> (f32 | bf16) re-layouts in ONE launch: the per-step operand copies of parameters whose stored layout no GEMM /
// stencil kernel reads directly (depthwise taps -> tap-major, k4s2 / k2s2 convolution weights -> [Cout][K], rows that are not 16-byte
// aligned in bf16).  ~170 launches of 5-8 us each per training step otherwise.
// tab: [num][10] int64 = src, dst, d0, d1, d2, s0, s1, s2, out_bf16, unused;  prefix: [num + 1] int64 block offsets; a block owns
// 1024 consecutive source elements of its entry:  dst[a*s0 + b*s1 + c*s2] = src[(a*d1 + b)*d2 + c].
__global__ __launch_bounds__(256) void permute3_multi_kernel(const long long* __restrict__ tab, const long long* __restrict__ prefix, int num) {
    int lo = 0, hi = num;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const long long* t = tab + (long)lo * 10;
    const float* src = reinterpret_cast<const float*>(t[0]);
    const int d1 = (int)t[3], d2 = (int)t[4];
    const long n = t[2] * t[3] * t[4], s0 = t[5], s1 = t[6], s2 = t[7];
    const bool bf = t[8] != 0;
    const long base = ((long long)blockIdx.x - prefix[lo]) * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long i = base + r * 256 + threadIdx.x;
        if (i >= n) break;
        const int c = (int)(i % d2); const long q = i / d2; const int b = (int)(q % d1); const long a = q / d1;
        const long o = a * s0 + b * s1 + c * s2;
        const float v = src[i];
        if (bf) TT<bf16raw>::st(reinterpret_cast<bf16raw*>(t[1]) + o, v);
        else reinterpret_cast<float*>(t[1])[o] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// LeFF depthwise 3x3 (net/utils/leff.py:104-111).  Pre- and post-activation tensors are both kept
// (h = pre, g = GELU(h)) so no erf is ever recomputed per tap:
//   fwd      : h2 = dwconv3x3(g1) + bias ;  g2 = GELU(h2)
//   bwd data : dh1 = GELU'(h1) * convT(dh2, w)
//   bwd wgt  : dw[c][tap] += sum_t g1[t+tap] dh2[t] ;  dbias[c] += sum_t dh2[t]
// Weights (and their gradient) are TAP-MAJOR f32 [9][C] so a lane's E weights of one tap are one vector load.
// A thread owns a strip of 8 consecutive pixels of one row and one 16-byte channel vector: the 72 weights are
// loaded once per strip and the 3 x 10 input vectors of the strip are each loaded once (3.75 loads per output
// instead of 9), with the 16-byte accesses of neighbouring threads contiguous along channels.
// ------------------------------------------------------------------------------------------------
constexpr int SX = 8;     // strip length (W is a multiple of 8 everywhere in the model)

// Raw 4-element vectors (16 B of f32 / 8 B of bf16) kept packed in registers until they are consumed.
template <typename T> struct Raw4;
template <> struct Raw4<float> {
    uint4 v;
    FW_MEM void load(const float* p) { v = *reinterpret_cast<const uint4*>(p); }
    FW_MEM void lds(const char* p) { v = *reinterpret_cast<const uint4*>(p); }
    FW_MEM void zero_unless(bool ok) { if (!ok) v = make_uint4(0, 0, 0, 0); }
    FW_MEM void launder() { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
    FW_MEM void unpack(float* f) const { unpack16<float>(v, f); }
};
template <> struct Raw4<bf16raw> {
    uint2 v;
    FW_MEM void load(const bf16raw* p) { v = *reinterpret_cast<const uint2*>(p); }
    FW_MEM void lds(const char* p) { v = *reinterpret_cast<const uint2*>(p); }
    FW_MEM void zero_unless(bool ok) { if (!ok) v = make_uint2(0, 0); }
    FW_MEM void launder() { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
    FW_MEM void unpack(float* f) const {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};

// MODE 0: forward (writes out = conv + bias and out2 = GELU(out));  MODE 1: data gradient (flipped taps, times GELU'(pre)).
// Branch-free: out-of-image taps are read from CLAMPED coordinates and zeroed by a select, so the 30 (+8) loads of a thread are
// independent instructions the compiler issues back to back -- with a branch around every edge load each load waited for the
// previous one (a chain of 30 memory latencies per thread: 2 TB/s instead of the ~5 TB/s this access pattern streams at).
template <typename T, int MODE, bool GIN = false>          // GIN: the input is the PRE-activation; GELU is applied as it is consumed
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const T* __restrict__ in, long ldi, const float* __restrict__ w, const float* __restrict__ bias,
                                    const T* __restrict__ pre, T* __restrict__ out, T* __restrict__ out2, long ldo, int B, int H, int W, int C) {
    constexpr int E = 4;                                   // channels per thread
    const int nv = C / E, ns = W / SX;
    const long total = (long)B * H * ns * nv;
    // workgroups are dealt round-robin to the 8 XCDs (8 private L2s): give every XCD one CONTIGUOUS eighth of the rows, so
    // that the three output rows sharing an input row meet it in the same L2 instead of fetching it on three XCDs
    const long lb = (gridDim.x % 8 == 0) ? (long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (long)blockIdx.x;
    for (long i = lb * TPB + threadIdx.x; i < total; i += gstride()) {
        const int v = (int)(i % nv); long t = i / nv;
        const int sx = (int)(t % ns); t /= ns;
        const int y = (int)(t % H); const long b = t / H;
        const int c0 = v * E, x0 = sx * SX;
        // ---- all loads first ----
        Raw4<T> raw[3][SX + 2];
        const bool left_ok = x0 > 0, right_ok = x0 + SX < W;
        const int xl = left_ok ? x0 - 1 : 0, xr = right_ok ? x0 + SX : W - 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int yy = y + ky - 1;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            const T* row = in + ((b * H + yy) * W) * ldi + c0;
            raw[ky][0].load(row + (long)xl * ldi);
#pragma unroll
            for (int cx = 0; cx < SX; ++cx) raw[ky][cx + 1].load(row + (long)(x0 + cx) * ldi);
            raw[ky][SX + 1].load(row + (long)xr * ldi);
        }
        const long tok0 = (b * H + y) * W + x0;
        Raw4<T> rpre[MODE == 1 ? SX : 1];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int o = 0; o < SX; ++o) rpre[o].load(pre + (tok0 + o) * ldi + c0);
        }
        float wr[E][9], acc[SX][E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wr[e][k] = w[(MODE == 0 ? k : 8 - k) * C + c0 + e];      // tap-major weights: contiguous per lane
            const float b0 = MODE == 0 ? bias[c0 + e] : 0.f;
#pragma unroll
            for (int o = 0; o < SX; ++o) acc[o][e] = b0;
        }
        // ---- then the arithmetic ----
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool row_ok = (y + ky - 1 >= 0) && (y + ky - 1 < H);
#pragma unroll
            for (int cx = -1; cx <= SX; ++cx) {
                Raw4<T> r = raw[ky][cx + 1];
                r.zero_unless(row_ok && (cx >= 0 || left_ok) && (cx < SX || right_ok));
                float f[E];
                r.unpack(f);
                if constexpr (GIN) {
#pragma unroll
                    for (int e = 0; e < E; ++e) f[e] = gelu_t<T>(f[e]);          // GELU(0) = 0: the zero padding stays zero
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int o = cx - kx + 1;
                    if (o >= 0 && o < SX) {
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[o][e] += f[e] * wr[e][ky * 3 + kx];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < SX; ++o) {
            if (MODE == 0) {
                stvec4<T>(out + (tok0 + o) * ldo + c0, acc[o]);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[o][e] = gelu_t<T>(acc[o][e]);
                stvec4<T>(out2 + (tok0 + o) * ldo + c0, acc[o]);
            } else {
                float hc[E];
                rpre[o].unpack(hc);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[o][e] *= gelu_grad_t<T>(hc[e]);
                stvec4<T>(out + (tok0 + o) * ldo + c0, acc[o]);
            }
        }
    }
}

// LDS-tiled form of the same stencil.  A workgroup owns 8 rows x 16 pixels x 64 channels: the 10 x 18 pixel halo tile is fetched
// ONCE from global memory (16-byte pieces, 128 contiguous bytes per pixel; 5.6 loads per thread instead of the strip kernel's 30,
// which was bound by the number of vector-memory instructions, not by latency: 4-pixel strips with more waves ran SLOWER),
// then every thread computes its 8-pixel x 4-channel strip from LDS (ds_read_b64, 100-cycle latency, no TA traffic).
// LDS layout [row][pixel][64 ch], 128-byte pixels, row stride 18 * 128 + 128 (= 128 mod 256: the two rows of a 32-lane half hit
// disjoint bank halves).  lane = (channel vector 0..15, row 0..7, strip 0..1).
constexpr int DT_TY = 8, DT_TX = 16, DT_CB = 64;
template <typename T, int MODE, bool GIN = false>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const T* __restrict__ in, long ldi, const float* __restrict__ w, const float* __restrict__ bias,
                                    const T* __restrict__ pre, T* __restrict__ out, T* __restrict__ out2, long ldo, int B, int H, int W, int C) {
    constexpr int SZ = TT<T>::SZ, E16 = TT<T>::E16;
    constexpr int PXB = DT_CB * SZ;                        // bytes per pixel in LDS
    constexpr int ROWB = (DT_TX + 2) * PXB + 128;          // row stride
    constexpr int CHUNKS = PXB / 16;                       // 16-byte pieces per pixel
    constexpr int NPIECE = (DT_TY + 2) * (DT_TX + 2) * CHUNKS;
    constexpr int NI = (NPIECE + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int ncb = (C + DT_CB - 1) / DT_CB, ntx = (W + DT_TX - 1) / DT_TX, nty = H / DT_TY;
    // contiguous eighth of the block order per XCD (vertical neighbours share halo rows in that L2)
    long bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int cb = (int)(bid % ncb); long t = bid / ncb;
    const int tx = (int)(t % ntx); t /= ntx;
    const int ty = (int)(t % nty); const long b = t / nty;
    if (b >= B) return;
    const int y0 = ty * DT_TY, x0 = tx * DT_TX, c0b = cb * DT_CB;
    // ---- halo tile -> LDS (two-phase, branch-free) ----
    uint4 r[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int ch = idx % CHUNKS, px = (idx / CHUNKS) % (DT_TX + 2), row = (idx / (CHUNKS * (DT_TX + 2))) % (DT_TY + 2);
        int yy = y0 + row - 1, xx = x0 + px - 1, cc = c0b + ch * E16;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx); cc = cc + E16 <= C ? cc : 0;
        r[i] = *reinterpret_cast<const uint4*>(in + ((b * H + yy) * W + xx) * ldi + cc);
    }
    // this thread's outputs
    const int cv = threadIdx.x & 15, ry = (threadIdx.x >> 4) & 7, sxl = threadIdx.x >> 7;
    const int c0 = c0b + cv * 4;
    const bool c_ok = c0 + 4 <= C;
    const int cw = c_ok ? c0 : 0;
    const int oy = y0 + ry, ox0 = x0 + sxl * 8;
    const long tok0 = (b * H + oy) * W + ox0;
    Raw4<T> rpre[MODE == 1 ? 8 : 1];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int xx = ox0 + o < W ? ox0 + o : W - 1;
            rpre[o].load(pre + ((b * H + oy) * W + xx) * ldi + cw);
        }
    }
    float wr[4][9], acc[8][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int k = 0; k < 9; ++k) wr[e][k] = w[(MODE == 0 ? k : 8 - k) * C + cw + e];
        const float b0 = MODE == 0 ? bias[cw + e] : 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o][e] = b0;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx >= NPIECE) continue;
        const int ch = idx % CHUNKS, px = (idx / CHUNKS) % (DT_TX + 2), row = idx / (CHUNKS * (DT_TX + 2));
        const int yy = y0 + row - 1, xx = x0 + px - 1, cc = c0b + ch * E16;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W && cc + E16 <= C;
        uint4 piece = r[i];
        if constexpr (GIN) {                                  // the pre-activation arrives: GELU once per halo element, on its way into LDS
            float f[E16];
            unpack16<T>(piece, f);
#pragma unroll
            for (int e = 0; e < E16; ++e) f[e] = gelu_t<T>(f[e]);
            piece = pack16<T>(f);
        }
        *reinterpret_cast<uint4*>(dsm + row * ROWB + px * PXB + ch * 16) = ok ? piece : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    // ---- stencil from LDS ----
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const char* rowp = dsm + (ry + ky) * ROWB + (sxl * 8) * PXB + cv * 4 * SZ;
#pragma unroll
        for (int cx = 0; cx < 10; ++cx) {
            Raw4<T> q;
            q.lds(rowp + cx * PXB);
            float f[4];
            q.unpack(f);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int o = cx - kx;                       // tile pixel cx is input x = ox0 + cx - 1
                if (o >= 0 && o < 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o][e] += f[e] * wr[e][ky * 3 + kx];
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (!c_ok || ox0 + o >= W) continue;
        if (MODE == 0) {
            stvec4<T>(out + (tok0 + o) * ldo + c0, acc[o]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] = gelu_t<T>(acc[o][e]);
            stvec4<T>(out2 + (tok0 + o) * ldo + c0, acc[o]);
        } else {
            float hc[4];
            rpre[o].unpack(hc);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] *= gelu_grad_t<T>(hc[e]);
            stvec4<T>(out + (tok0 + o) * ldo + c0, acc[o]);
        }
    }
}

// weight / bias gradient.  block = 8 strip lanes x 32 channel vectors (4 channels each: 256 contiguous bytes per pixel); a thread
// walks NSTRIP consecutive strips with its 10 x 4 partial sums in registers; partials are folded over the block (one shuffle +
// LDS) before ONE atomic per word.  Loads are branch-free (clamped coordinates, zeroed by select) so they issue back to back.
// INPUT-centric: a thread owns 8 INPUT pixels (the activation g1, or -- GIN -- the pre-activation h1, whose GELU is then evaluated
// exactly once per element) and gathers the 3 x 10 halo of the OUTPUT gradient:
//     dw[tap (ky, kx)] += g1[y][x] * dh2[y - ky + 1][x - kx + 1]        (dh2 = 0 outside the image)
// -- the same sum as the output-centric form, which would need GELU on its 3.75x re-read input halo.
constexpr int WG_VL = 32, WG_SL = 8;
template <typename T, bool GIN>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ dh2, long ldg, const T* __restrict__ g1, long ld1,
                                                           float* __restrict__ dw, float* __restrict__ dbias, int B, int H, int W, int C, int NSTRIP) {
    constexpr int E = 4;                                   // channels per thread
    __shared__ float red[WG_VL * E * 10];
    const int nv = C / E, ns = W / SX;
    const int nvg = (nv + WG_VL - 1) / WG_VL;
    const long nstrips = (long)B * H * ns;
    const int vl = threadIdx.x & (WG_VL - 1), sl = threadIdx.x / WG_VL;
    const int v = (blockIdx.x % nvg) * WG_VL + vl;
    const long s0 = ((long)(blockIdx.x / nvg) * WG_SL + sl) * NSTRIP;
    const bool live = v < nv;
    const int c0 = (live ? v : 0) * E;
    for (int i = threadIdx.x; i < WG_VL * E * 10; i += 256) red[i] = 0.f;
    float gw[10][E];
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) gw[k][e] = 0.f;
    if (live)
        for (long s = s0; s < nstrips && s < s0 + NSTRIP; ++s) {
            const int sx = (int)(s % ns); const int y = (int)((s / ns) % H); const long b = s / ((long)ns * H);
            const int x0 = sx * SX;
            const bool left_ok = x0 > 0, right_ok = x0 + SX < W;
            const int xl = left_ok ? x0 - 1 : 0, xr = right_ok ? x0 + SX : W - 1;
            Raw4<T> rg[SX], raw[3][SX + 2];
            const T* grow = g1 + ((b * H + y) * W + x0) * ld1 + c0;
#pragma unroll
            for (int o = 0; o < SX; ++o) rg[o].load(grow + (long)o * ld1);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int yy = y - ky + 1;
                yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                const T* row = dh2 + ((b * H + yy) * W) * ldg + c0;
                raw[ky][0].load(row + (long)xl * ldg);
#pragma unroll
                for (int cx = 0; cx < SX; ++cx) raw[ky][cx + 1].load(row + (long)(x0 + cx) * ldg);
                raw[ky][SX + 1].load(row + (long)xr * ldg);
            }
            float g[SX][E];
#pragma unroll
            for (int o = 0; o < SX; ++o) {
                rg[o].unpack(g[o]);
                if constexpr (GIN) {
#pragma unroll
                    for (int e = 0; e < E; ++e) g[o][e] = gelu_t<T>(g[o][e]);
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const bool row_ok = (y - ky + 1 >= 0) && (y - ky + 1 < H);
#pragma unroll
                for (int j = 0; j < SX + 2; ++j) {               // dh2 at column x0 - 1 + j
                    Raw4<T> r = raw[ky][j];
                    r.zero_unless(row_ok && (j >= 1 || left_ok) && (j <= SX || right_ok));
                    float d[E];
                    r.unpack(d);
                    if (ky == 1 && j >= 1 && j <= SX) {
#pragma unroll
                        for (int e = 0; e < E; ++e) gw[9][e] += d[e];
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int i = j + kx - 2;                // input pixel of the strip: j = i - kx + 2
                        if (i >= 0 && i < SX) {
#pragma unroll
                            for (int e = 0; e < E; ++e) gw[ky * 3 + kx][e] += g[i][e] * d[e];
                        }
                    }
                }
            }
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float sum = gw[k][e];
            sum += __shfl_xor(sum, 32, 64);                 // the wave's two strip lanes
            if ((threadIdx.x & 63) < WG_VL) atomicAdd(&red[(vl * E + e) * 10 + k], sum);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < WG_VL * E * 10; i += 256) {
        const int k = i % 10, ce = i / 10;
        const int c = (blockIdx.x % nvg) * WG_VL * E + ce;
        if (c < C) {
            if (k < 9) atomicAdd(dw + (long)c * 9 + k, red[i]);                       // the parameter's own [C][3][3] layout
            else atomicAdd(dbias + c, red[i]);
        }
    }
}

// Forward stencil as a PERSISTENT tile loop (the scheme of dwconv_bwd_fused_kernel below): a workgroup walks NT vertically consecutive
// 8 x 16 x 64-channel tiles, the next tile's halo pieces are in flight (registers) while the current one is computed from LDS, the
// per-piece address arithmetic is done once per workgroup, the 9 x 64 weights live in LDS (one float4 per tap and thread).
template <typename T, bool GIN>
__global__ __launch_bounds__(256, 3) void dwconv_fwd_pipe_kernel(const T* __restrict__ in, long ldi, const float* __restrict__ w, const float* __restrict__ bias,
                                                              T* __restrict__ out, T* __restrict__ out2, long ldo, int B, int H, int W, int C, int NT) {
    constexpr int SZ = TT<T>::SZ, E16 = TT<T>::E16;
    constexpr int PXB = DT_CB * SZ, ROWB = (DT_TX + 2) * PXB + 128, CHUNKS = PXB / 16;
    constexpr int NPIECE = (DT_TY + 2) * (DT_TX + 2) * CHUNKS, NI = (NPIECE + 255) / 256;
    constexpr int TILEB = (DT_TY + 2) * ROWB;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int ncb = (C + DT_CB - 1) / DT_CB, ntx = (W + DT_TX - 1) / DT_TX, nty = H / DT_TY;
    const int ncol = ncb * ntx;
    const long nrow = (long)B * nty, nchunk = (nrow + NT - 1) / NT;
    long bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int col = (int)(bid % ncol);
    const long chunk = bid / ncol;
    if (chunk >= nchunk) return;
    const int cb = col % ncb, tx = col / ncb;
    const long s0 = chunk * NT, s1 = s0 + NT < nrow ? s0 + NT : nrow;
    const int x0 = tx * DT_TX, c0b = cb * DT_CB;
    int ppack[NI], pxo[NI];
    unsigned pokm = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int ch = idx % CHUNKS, px = (idx / CHUNKS) % (DT_TX + 2), row = (idx / (CHUNKS * (DT_TX + 2))) % (DT_TY + 2);
        int xx = x0 + px - 1, cc = c0b + ch * E16;
        if (idx < NPIECE && xx >= 0 && xx < W && cc + E16 <= C) pokm |= 1u << i;
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx); cc = cc + E16 <= C ? cc : 0;
        pxo[i] = xx * (int)ldi + cc; ppack[i] = (row * ROWB + px * PXB + ch * 16) | (row << 24);
    }
    const int cv = threadIdx.x & 15, ry = (threadIdx.x >> 4) & 7, sxl = threadIdx.x >> 7;
    const int c0 = c0b + cv * 4;
    const bool c_ok = c0 + 4 <= C;
    const int cw = c_ok ? c0 : 0;
    const int ox0 = x0 + sxl * 8;
    float* wl = reinterpret_cast<float*>(dsm + TILEB);                  // [9][64]
    for (int i = threadIdx.x; i < 9 * DT_CB; i += 256) {
        const int c = c0b + (i % DT_CB);
        wl[i] = c < C ? w[(i / DT_CB) * C + c] : 0.f;
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + cw);
    const int rowelems = W * (int)ldi;
    uint4 r[NI];
    auto issue = [&](long s) {
        const T* base = in + (s / nty) * H * (long)rowelems;
        const int y0 = (int)(s % nty) * DT_TY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int yy = y0 + (ppack[i] >> 24) - 1;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            r[i] = *reinterpret_cast<const uint4*>(base + (yy * rowelems + pxo[i]));
        }
    };
    issue(s0);
    for (long s = s0; s < s1; ++s) {
        const long b = s / nty;
        const int y0 = (int)(s % nty) * DT_TY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (threadIdx.x + i * 256 >= NPIECE) continue;
            const int yy = y0 + (ppack[i] >> 24) - 1;
            uint4 piece = r[i];
            if constexpr (GIN) {                              // the pre-activation arrives: GELU once per halo element, on its way into LDS
                float f[E16];
                unpack16<T>(piece, f);
#pragma unroll
                for (int e = 0; e < E16; ++e) f[e] = gelu_t<T>(f[e]);
                piece = pack16<T>(f);
            }
            *reinterpret_cast<uint4*>(dsm + (ppack[i] & 0xffffff)) = (((pokm >> i) & 1) && yy >= 0 && yy < H) ? piece : make_uint4(0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                // one piece at a time: the scheduler otherwise unpacks all six at once (212 VGPRs)
        }
        __syncthreads();
        if (s + 1 < s1) issue(s + 1);                         // in flight while this tile is computed
        float acc[8][4];
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] = b4[e];
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const char* rowp = dsm + (ry + ky) * ROWB + (sxl * 8) * PXB + cv * 4 * SZ;
            f32x4 wk[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wk[kx] = *reinterpret_cast<const f32x4*>(wl + (ky * 3 + kx) * DT_CB + cv * 4);
#pragma unroll
            for (int cx = 0; cx < 10; ++cx) {
                Raw4<T> q;
                q.lds(rowp + cx * PXB);
                float f[4];
                q.unpack(f);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int o = cx - kx;
                    if (o >= 0 && o < 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[o][e] += f[e] * wk[kx][e];
                    }
                }
            }
        }
        const long tok0 = ((b * H + y0 + ry) * W + ox0) * ldo + c0;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const bool ok = c_ok && ox0 + o < W;
            if (ok) stvec4<T>(out + tok0 + (long)o * ldo, acc[o]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] = gelu_t<T>(acc[o][e]);
            if (ok) stvec4<T>(out2 + tok0 + (long)o * ldo, acc[o]);
        }
        __syncthreads();
    }
}

// Whole backward of the depthwise 3x3 in ONE pass over its operands (dh2 and h1 read once, dh1 written once; the two-kernel form read
// dh2 and h1 twice -- 5 tensor passes instead of 3 -- and evaluated the 30-tap gather twice).  Both gradients gather the SAME
// neighbourhood of dh2 around an input pixel p:
//     dh1[p]      = GELU'(h1[p]) * sum_tap w[tap] dh2[p - tap + 1]
//     dw[tap]    += GELU(h1[p]) * dh2[p - tap + 1]            dbias += dh2[p]
// so the data-gradient tile kernel's stencil loop carries one more FMA per tap.  A workgroup is PERSISTENT over NT vertically
// consecutive 8 x 16 x 64-channel tiles: its 10 x 4 weight-gradient partials stay in registers across tiles and end in one fold
// (wave shuffles, LDS, then 640 atomics per workgroup instead of per tile), and the next tile's halo pieces / h1 vectors are in
// flight (registers) while the current tile is computed from LDS -- the one-shot tile kernel had no overlap inside a workgroup.
template <typename T>
__global__ __launch_bounds__(256, 2) void dwconv_bwd_fused_kernel(const T* __restrict__ dh2, long ldg, const T* __restrict__ h1, const float* __restrict__ w,
                                                               T* __restrict__ dh1, long ldo, float* __restrict__ dw, float* __restrict__ dbias,
                                                               int B, int H, int W, int C, int NT) {
    constexpr int SZ = TT<T>::SZ, E16 = TT<T>::E16;
    constexpr int PXB = DT_CB * SZ, ROWB = (DT_TX + 2) * PXB + 128, CHUNKS = PXB / 16;
    constexpr int NPIECE = (DT_TY + 2) * (DT_TX + 2) * CHUNKS, NI = (NPIECE + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int ncb = (C + DT_CB - 1) / DT_CB, ntx = (W + DT_TX - 1) / DT_TX, nty = H / DT_TY;
    const int ncol = ncb * ntx;
    const long nrow = (long)B * nty, nchunk = (nrow + NT - 1) / NT;
    long bid = blockIdx.x;                                   // contiguous eighth of the block order per XCD: horizontal neighbours share halo columns in that L2
    if ((gridDim.x & 7) == 0) bid = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int col = (int)(bid % ncol);
    const long chunk = bid / ncol;
    if (chunk >= nchunk) return;
    const int cb = col % ncb, tx = col / ncb;
    const long s0 = chunk * NT, s1 = s0 + NT < nrow ? s0 + NT : nrow;
    const int x0 = tx * DT_TX, c0b = cb * DT_CB;
    // ---- tile-independent part of the halo pieces (32-bit offsets: a tile row base is uniform, the rest of an address fits an int) ----
    int ppack[NI], pxo[NI];                                  // LDS offset | halo row << 24;  element offset of (clamped x, channel chunk)
    unsigned pokm = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int ch = idx % CHUNKS, px = (idx / CHUNKS) % (DT_TX + 2), row = (idx / (CHUNKS * (DT_TX + 2))) % (DT_TY + 2);
        int xx = x0 + px - 1, cc = c0b + ch * E16;
        if (idx < NPIECE && xx >= 0 && xx < W && cc + E16 <= C) pokm |= 1u << i;
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx); cc = cc + E16 <= C ? cc : 0;
        pxo[i] = xx * (int)ldg + cc; ppack[i] = (row * ROWB + px * PXB + ch * 16) | (row << 24);
    }
    const int cv = threadIdx.x & 15, ry = (threadIdx.x >> 4) & 7, sxl = threadIdx.x >> 7;
    const int c0 = c0b + cv * 4;
    const bool c_ok = c0 + 4 <= C;
    const int cw = c_ok ? c0 : 0;
    const int ox0 = x0 + sxl * 8;
    // LDS beyond the halo tile: the 9 x 64 flipped weights of this channel block (read back as one float4 per tap: 36 registers less
    // than holding them), and the weight-gradient partials, one private slot per (value, thread) -- [40][256] floats, conflict-free.
    constexpr int TILEB = (DT_TY + 2) * ROWB;
    float* wl = reinterpret_cast<float*>(dsm + TILEB);                  // [9][64]
    float* part = wl + 9 * DT_CB;                                       // [40][256]
    for (int i = threadIdx.x; i < 9 * DT_CB; i += 256) {
        const int k = i / DT_CB, c = c0b + (i % DT_CB);
        wl[i] = c < C ? w[(8 - k) * C + c] : 0.f;                       // flipped taps, as in the data-gradient kernels
    }
#pragma unroll
    for (int j = 0; j < 40; ++j) part[j * 256 + threadIdx.x] = 0.f;
    float gb[4] = {0.f, 0.f, 0.f, 0.f};                                 // bias-gradient partial
    uint4 r[NI];
    Raw4<T> pre[8];
    const int rowelems = W * (int)ldg;                        // elements per image row (< 2^31 / H: checked by the host)
    auto issue_halo = [&](long s) {                           // the dh2 halo pieces of tile row s (branch-free: clamped coordinates)
        const T* base = dh2 + (s / nty) * H * (long)rowelems;
        const int y0 = (int)(s % nty) * DT_TY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int yy = y0 + (ppack[i] >> 24) - 1;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            r[i] = *reinterpret_cast<const uint4*>(base + (yy * rowelems + pxo[i]));
        }
    };
    auto issue_pre = [&](long s) {                            // this thread's 8 h1 vectors of tile row s
        const T* base = h1 + (s / nty) * H * (long)rowelems;
        const int yoff = ((int)(s % nty) * DT_TY + ry) * rowelems + cw;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int xx = ox0 + o < W ? ox0 + o : W - 1;
            pre[o].load(base + (yoff + xx * (int)ldg));
        }
    };
    issue_halo(s0);
    issue_pre(s0);
    for (long s = s0; s < s1; ++s) {
        const long b = s / nty;
        const int y0 = (int)(s % nty) * DT_TY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (threadIdx.x + i * 256 >= NPIECE) continue;
            const int yy = y0 + (ppack[i] >> 24) - 1;
            *reinterpret_cast<uint4*>(dsm + (ppack[i] & 0xffffff)) = (((pokm >> i) & 1) && yy >= 0 && yy < H) ? r[i] : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        float g[8][4], acc[8][4];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float hc[4];
            pre[o].unpack(hc);
            const bool in = ox0 + o < W;                     // a strip pixel beyond the image edge contributes nothing
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[o][e] = in ? gelu_t<T>(hc[e]) : 0.f; acc[o][e] = 0.f; }
            if (o & 1) __builtin_amdgcn_sched_barrier(0);       // eight polynomials at a time, not thirty-two (their temporaries are registers)
        }
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            const char* rowp = dsm + (ry + ky) * ROWB + (sxl * 8) * PXB + cv * 4 * SZ;
            f32x4 wk[3];
            float gk[3][4];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                wk[kx] = *reinterpret_cast<const f32x4*>(wl + (ky * 3 + kx) * DT_CB + cv * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) gk[kx][e] = 0.f;
            }
            const float bsel = ky == 1 ? 1.f : 0.f;             // the centre row carries the bias gradient
#pragma unroll
            for (int cx = 0; cx < 10; ++cx) {
                Raw4<T> q;
                q.lds(rowp + cx * PXB);
                float f[4];
                q.unpack(f);
                if (cx >= 1 && cx <= 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) gb[e] += bsel * f[e];
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int o = cx - kx;                   // tile pixel cx is x = ox0 + cx - 1
                    if (o >= 0 && o < 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[o][e] += f[e] * wk[kx][e];
                            gk[kx][e] += f[e] * g[o][e];
                        }
                    }
                }
                if (cx == 4) __builtin_amdgcn_sched_barrier(0);   // two batches of five LDS reads per row, not all thirty of a tile at once
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[((ky * 3 + kx) * 4 + e) * 256 + threadIdx.x] += gk[kx][e];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < s1) issue_halo(s + 1);                    // the next halo flies under the epilogue (its registers are the ones g[] just left)
        __builtin_amdgcn_sched_barrier(0);
        T* orow = dh1 + ((b * H + y0 + ry) * W + ox0) * ldo + c0;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float hc[4];
            Raw4<T> again = pre[o];
            again.launder();                                  // a fresh unpack here: otherwise the 32 floats unpacked for GELU stay live across the stencil
            again.unpack(hc);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] *= gelu_grad_t<T>(hc[e]);
            if (c_ok && ox0 + o < W) stvec4<T>(orow + (long)o * ldo, acc[o]);
            if (o & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < s1) issue_pre(s + 1);                     // h1 of the next tile: in flight across the barrier and the LDS staging
        __syncthreads();                                      // every wave is done with the tile before the next one overwrites it
    }
    // ---- fold the partials of the 16 threads that share a channel vector: ry bits inside the wave, then the 4 waves through LDS ----
    float* red = reinterpret_cast<float*>(dsm);               // the halo tile is free now (the loop ended in a barrier)
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 40; ++j) {
        float v = j < 36 ? part[j * 256 + threadIdx.x] : gb[j - 36];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if ((threadIdx.x & 63) < 16) red[(wave * 16 + cv) * 40 + j] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 40; i += 256) {
        const int cvv = i / 40, j = i % 40, k = j >> 2, e = j & 3;
        const float sum = red[cvv * 40 + j] + red[(16 + cvv) * 40 + j] + red[(32 + cvv) * 40 + j] + red[(48 + cvv) * 40 + j];
        const int c = c0b + cvv * 4 + e;
        if (c < C) {
            if (k < 9) atomicAdd(dw + (long)c * 9 + (8 - k), sum);                    // gw[k] pairs with the flipped tap 8 - k
            else atomicAdd(dbias + c, sum);
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
// narrow matrices (cols <= 1024, whole float4 columns): all 256 threads busy -- thread = (float4 column, row lane), 16-byte loads, the
// row lanes folded through LDS, one atomic per column and block (the plain kernel above keeps cols / 256 of its threads busy: 187 us for
// the 28-column bias gradient of the last upsampling layer at 256 x 256).
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, long rows, int cols, int rows_per_block) {
    __shared__ f32x4 red[TPB];
    const int nv = cols >> 2, rl = TPB / nv;
    const int v = threadIdx.x % nv, lane = threadIdx.x / nv;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (lane < rl) {
#pragma unroll 4
        for (long r = r0 + lane; r < r1; r += rl) s += *reinterpret_cast<const f32x4*>(x + r * ldx + 4 * v);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < nv) {
        f32x4 t = red[threadIdx.x];
        for (int j = 1; j < rl; ++j) t += red[j * nv + threadIdx.x];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(out + 4 * threadIdx.x + e, t[e]);
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
__global__ __launch_bounds__(256) void inproj_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, long ldo, int B, int H, int W, int C, float slope) {
    extern __shared__ __attribute__((aligned(16))) float sm[];         // tap-major weights [27][C] + bias [C]
    for (int i = threadIdx.x; i < C * 27; i += 256) sm[(i % 27) * C + i / 27] = w[i];
    for (int i = threadIdx.x; i < C; i += 256) sm[27 * C + i] = bias[i];
    __syncthreads();
    const int c4n = C >> 2;
    const long total = (long)B * H * W * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; const long tok = i / c4n;
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        f32x4 acc = *reinterpret_cast<const f32x4*>(sm + 27 * C + c);
        // branch-free: the 27 taps are read from clamped coordinates (independent loads, issued together) and zeroed by a select --
        // a load under a lane-varying branch is waited for before the next one is issued (137 us per launch with the branches)
        float px[27];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int yy = y + ky - 1; yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int xx = x + kx - 1; xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                    px[ci * 9 + ky * 3 + kx] = img[((b * 3 + ci) * H + yy) * W + xx];
                }
            }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool yok = y + ky - 1 >= 0 && y + ky - 1 < H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bool ok = yok && x + kx - 1 >= 0 && x + kx - 1 < W;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
                    acc += *reinterpret_cast<const f32x4*>(sm + (ci * 9 + ky * 3 + kx) * C + c) * (ok ? px[ci * 9 + ky * 3 + kx] : 0.f);
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
#pragma unroll
        for (int t = 0; t < 27; ++t) g[t] = 0.f;
        for (long tok = st * STRIPE; tok < ntok && tok < (st + 1) * STRIPE; ++tok) {
            const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
            // branch-free: the 27 image taps are read from clamped coordinates (independent loads, issued together) and
            // zeroed by a 0/1 factor -- a load under a lane-varying branch is waited for before the next one is issued
            float px[27];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int yy = y + ky - 1; yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        int xx = x + kx - 1; xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                        px[ci * 9 + ky * 3 + kx] = img[((b * 3 + ci) * H + yy) * W + xx];
                    }
                }
            float d = dy[tok * ldy + c];
            const float o = out[tok * ldo + c];
            d = o <= 0.f ? d * slope : d;
            gb += d;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float dyk = (y + ky - 1 >= 0 && y + ky - 1 < H) ? d : 0.f;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float dk = (x + kx - 1 >= 0 && x + kx - 1 < W) ? dyk : 0.f;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) g[ci * 9 + ky * 3 + kx] += dk * px[ci * 9 + ky * 3 + kx];
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
__global__ __launch_bounds__(256) void outproj_fwd_kernel(const float* __restrict__ fea, long ldf, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ img, float* __restrict__ out, int B, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];         // weights as [tap][co][C]
    for (int i = threadIdx.x; i < 3 * C * 9; i += 256) { const int tap = i % 9, cc = (i / 9) % C, co = i / (9 * C); sm[(tap * 3 + co) * C + cc] = w[i]; }
    __syncthreads();
    const long total = (long)B * H * W;
    for (long tok = gtid(); tok < total; tok += gstride()) {
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                if (xx < 0 || xx >= W) continue;
                const float* f = fea + ((b * H + yy) * W + xx) * ldf;
                const float* wt = sm + (ky * 3 + kx) * 3 * C;
                for (int c = 0; c < C; c += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(f + c);
                    a0 += v * *reinterpret_cast<const f32x4*>(wt + c);
                    a1 += v * *reinterpret_cast<const f32x4*>(wt + C + c);
                    a2 += v * *reinterpret_cast<const f32x4*>(wt + 2 * C + c);
                }
            }
        }
        const long p = (b * 3 * H + y) * W + x;
        const long hw = (long)H * W;
        out[p] = bias[0] + a0[0] + a0[1] + a0[2] + a0[3] + (img ? img[p] : 0.f);
        out[p + hw] = bias[1] + a1[0] + a1[1] + a1[2] + a1[3] + (img ? img[p + hw] : 0.f);
        out[p + 2 * hw] = bias[2] + a2[0] + a2[1] + a2[2] + a2[3] + (img ? img[p + 2 * hw] : 0.f);
    }
}
// dfea[tok][c] = sum_{co,tap} dout[co][tok - tap] w[co][c][tap]
__global__ __launch_bounds__(256) void outproj_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w, float* __restrict__ dfea, long ldf,
                                                               int B, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];         // weights as [tap][co][C]
    for (int i = threadIdx.x; i < 3 * C * 9; i += 256) { const int tap = i % 9, cc = (i / 9) % C, co = i / (9 * C); sm[(tap * 3 + co) * C + cc] = w[i]; }
    __syncthreads();
    const int c4n = C >> 2;
    const long total = (long)B * H * W * c4n;
    for (long i = gtid(); i < total; i += gstride()) {
        const int c = (int)(i % c4n) * 4; const long tok = i / c4n;
        const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yo = y - ky + 1;
            if (yo < 0 || yo >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xo = x - kx + 1;
                if (xo < 0 || xo >= W) continue;
#pragma unroll
                for (int co = 0; co < 3; ++co)
                    acc += *reinterpret_cast<const f32x4*>(sm + ((ky * 3 + kx) * 3 + co) * C + c) * dout[((b * 3 + co) * H + yo) * W + xo];
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
#pragma unroll
        for (int t = 0; t < 27; ++t) g[t] = 0.f;
        for (long tok = st * STRIPE; tok < ntok && tok < (st + 1) * STRIPE; ++tok) {
            const int x = (int)(tok % W); const int y = (int)((tok / W) % H); const long b = tok / ((long)W * H);
            float d[3], f[9];                              // branch-free: 9 clamped feature taps issued together, zeroed by a 0/1 factor
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int yy = y + ky - 1; yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int xx = x + kx - 1; xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                    f[ky * 3 + kx] = fea[((b * H + yy) * W + xx) * ldf + c];
                }
            }
#pragma unroll
            for (int co = 0; co < 3; ++co) { d[co] = dout[((b * 3 + co) * H + y) * W + x]; gb[co] += d[co]; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float fv = (y + ky - 1 >= 0 && y + ky - 1 < H && x + kx - 1 >= 0 && x + kx - 1 < W) ? f[ky * 3 + kx] : 0.f;
#pragma unroll
                    for (int co = 0; co < 3; ++co) g[co * 9 + ky * 3 + kx] += d[co] * fv;
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
// hyper (device, f32[4]): lr, beta1^t, beta2^t, t ; advanced by adam_tick_kernel so that a captured
// HIP graph replays the right bias correction and learning rate every step.
__global__ void adam_tick_kernel(float* __restrict__ hyper, float b1, float b2) { hyper[1] *= b1; hyper[2] *= b2; hyper[3] += 1.f; }
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
// dst[i] (+)= sum_z slab[z*zstride + i]: the reduction step of a split-K GEMM (replaces thousands of same-address atomics).
// Two destination segments in one launch: [0, n) -> dst, [off2, off2 + n2) -> dst2 (weight and bias gradient of one GEMM).
// One wave = 64 consecutive 16-byte columns (1 KB per slab row); the 4 waves of a block take every 4th slab row of the
// block's z range (independent, unrolled loads), are folded through LDS, and wave 0 writes the result.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nz, long n, long zstride, float* __restrict__ dst,
                                                          int accumulate, int zper, float* __restrict__ dst2, long off2, long n2) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long i4 = (long)blockIdx.x * 64 + lane;
    const long end = dst2 ? off2 + n2 : n;
    const bool live = i4 * 4 < end;
    const int z0 = blockIdx.y * zper, z1 = min(nz, z0 + zper);
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* p = slab + i4 * 4;
        int z = z0 + zl;
        for (; z + 12 < z1; z += 16) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p + (long)z * zstride);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(p + (long)(z + 4) * zstride);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(p + (long)(z + 8) * zstride);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(p + (long)(z + 12) * zstride);
            s += (a0 + a1) + (a2 + a3);
        }
        for (; z < z1; z += 4) s += *reinterpret_cast<const f32x4*>(p + (long)z * zstride);   // rows are padded to 4
    }
    if (zl > 0) red[zl - 1][lane] = s;
    __syncthreads();
    if (zl > 0 || !live) return;
    s += red[0][lane] + red[1][lane] + red[2][lane];
    for (int e = 0; e < 4; ++e) {
        const long i = i4 * 4 + e;
        float* d = nullptr;
        if (i < n) d = dst + i;
        else if (dst2 && i >= off2 && i < off2 + n2) d = dst2 + (i - off2);
        if (!d) continue;
        if (gridDim.y > 1) atomicAdd(d, s[e]);        // <= nz/zper adders per address
        else if (accumulate) *d += s[e];
        else *d = s[e];
    }
}
// Many slabs in one launch (every weight-gradient / LayerNorm partial of a backward pass, folded once at its end).
// tab: [num][12] int64 = slab, dst, dst2, n, zstride, off2, n2, nz, zper, nchunks, upw, atomics;  prefix: [num + 1] int64 block
// offsets.  Work unit = (64 consecutive 16-byte columns, one z range of zper slab rows); a WAVE owns upw consecutive units and
// sums each with 8 independent loads in flight -- no LDS, no barrier; the host sizes upw so that a wave moves >= ~32 KB.
// Every entry accumulates into dst / dst2: with atomics when its z range is split or its destination is shared with another entry,
// else by plain 16-byte read-modify-write (same-address float atomics run at ~1.3 TB/s, a quarter of the streaming rate).
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const long long* __restrict__ tab, const long long* __restrict__ prefix, int num) {
    int lo = 0, hi = num;                                  // entry e with prefix[e] <= blockIdx.x < prefix[e + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const long long* t = tab + (long)lo * 12;
    const float* slab = reinterpret_cast<const float*>(t[0]);
    float* dst = reinterpret_cast<float*>(t[1]);
    float* dst2 = reinterpret_cast<float*>(t[2]);
    const long n = t[3], zstride = t[4], off2 = t[5], n2 = t[6];
    const int nz = (int)t[7], zper = (int)t[8];
    const long nchunks = t[9];
    const int upw = (int)t[10];
    const bool use_atomics = t[11] != 0;                   // z-split entry, or a destination that another entry writes too
    const long splits = (nz + zper - 1) / zper, units = nchunks * splits;
    const long gw = ((long long)blockIdx.x - prefix[lo]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long end = dst2 ? off2 + n2 : n;
    for (long u = gw * upw; u < units && u < (gw + 1) * upw; ++u) {
        const long c = u % nchunks; const int sp = (int)(u / nchunks);
        const long i4 = c * 64 + lane;
        if (i4 * 4 >= end) continue;
        const int z0 = sp * zper, z1 = min(nz, z0 + zper);
        const float* p = slab + i4 * 4;
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
        int z = z0;
        for (; z + 7 < z1; z += 8) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p + (long)z * zstride);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(p + (long)(z + 1) * zstride);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(p + (long)(z + 2) * zstride);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(p + (long)(z + 3) * zstride);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(p + (long)(z + 4) * zstride);
            const f32x4 a5 = *reinterpret_cast<const f32x4*>(p + (long)(z + 5) * zstride);
            const f32x4 a6 = *reinterpret_cast<const f32x4*>(p + (long)(z + 6) * zstride);
            const f32x4 a7 = *reinterpret_cast<const f32x4*>(p + (long)(z + 7) * zstride);
            s0 += (a0 + a1) + (a2 + a3);
            s1 += (a4 + a5) + (a6 + a7);
        }
        for (; z < z1; ++z) s0 += *reinterpret_cast<const f32x4*>(p + (long)z * zstride);   // rows are padded to 4
        s0 += s1;
        const long i0 = i4 * 4;
        if (!use_atomics && i0 + 4 <= n) {                   // sole producer of these words: plain 16-byte read-modify-write
            f32x4* d = reinterpret_cast<f32x4*>(dst + i0);
            *d = *d + s0;
            continue;
        }
        if (!use_atomics && dst2 && i0 >= off2 && i0 + 4 <= off2 + n2 && ((off2 & 3) == 0)) {
            f32x4* d = reinterpret_cast<f32x4*>(dst2 + (i0 - off2));
            *d = *d + s0;
            continue;
        }
        for (int e = 0; e < 4; ++e) {
            const long i = i0 + e;
            float* d = nullptr;
            if (i < n) d = dst + i;
            else if (dst2 && i >= off2 && i < off2 + n2) d = dst2 + (i - off2);
            if (d) atomicAdd(d, s0[e]);
        }
    }
}
__global__ void fill_kernel(float* __restrict__ p, long n, float v) {
    for (long i = gtid(); i < n; i += gstride()) p[i] = v;
}

}  // namespace

#define ST ((hipStream_t)stream)
static int dw_tiled() { static const int v = getenv("FW_DWCONV_TILED") ? atoi(getenv("FW_DWCONV_TILED")) : 1; return v; }
static long dw_tiled_min() { static const long v = getenv("FW_DWCONV_TILED_MIN") ? atol(getenv("FW_DWCONV_TILED_MIN")) : 10000000L; return v; }   // elements B*H*W*C from which the LDS-tiled form is used: 272.8 images/s at 80 M (stage 0 only), 273.7 at 20 M, 274.6 at 10 M, 274.0 at 1 M
template <typename T, int MODE, bool GIN = false>
static int dwconv_tile_launch(const T* in, long ldi, const float* w, const float* bias, const T* pre, T* out, T* out2, long ldo, int B, int H, int W,
                              int C, hipStream_t st) {
    const size_t lds = (size_t)(DT_TY + 2) * ((DT_TX + 2) * DT_CB * sizeof(T) + 128);
    FW_SET_LDS_ONCE((dwconv_tile_kernel<T, MODE, GIN>), lds);
    long nb = (long)B * (H / DT_TY) * ((W + DT_TX - 1) / DT_TX) * ((C + DT_CB - 1) / DT_CB);
    nb = (nb + 7) / 8 * 8;
    hipLaunchKernelGGL((dwconv_tile_kernel<T, MODE, GIN>), dim3((unsigned)nb), dim3(256), lds, st, in, ldi, w, bias, pre, out, out2, ldo, B, H, W, C);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
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
extern "C" int fw_dwconv_fwd(int dtype, const void* g1, long ld1, int in_gelu, const float* w, const float* bias, void* h2, void* g2, long ld2,
                             int B, int H, int W, int C, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(g1 && w && bias && h2 && g2 && C % e == 0 && ld1 % e == 0 && ld2 % e == 0 && W % SX == 0);
    const long n = (long)B * H * (W / SX) * (C / 4);
    // the LDS-tiled form wins where the tensors outgrow the 256 MB infinity cache (stage-0 layers); below that the strips are as fast
    static const int piped = getenv("FW_DWCONV_PIPE_FWD") ? atoi(getenv("FW_DWCONV_PIPE_FWD")) : 1;
    if (piped && dw_tiled() && H % DT_TY == 0 && (long)B * H * W * C >= dw_tiled_min() && (long)H * W * ld1 < (1L << 31)) {
        static const long want = getenv("FW_DWCONV_PIPE_WGS") ? atol(getenv("FW_DWCONV_PIPE_WGS")) : 2048;
        const long ncol = (long)((C + DT_CB - 1) / DT_CB) * ((W + DT_TX - 1) / DT_TX), nrow = (long)B * (H / DT_TY);
        long NT = ncol * nrow / want;
        NT = NT < 1 ? 1 : (NT > 32 ? 32 : NT);
        long nb = ncol * ((nrow + NT - 1) / NT);
        nb = (nb + 7) / 8 * 8;
        const size_t esz = dtype == FW_DT_BF16 ? 2 : 4;
        const size_t lds = (size_t)(DT_TY + 2) * ((DT_TX + 2) * DT_CB * esz + 128) + 9 * DT_CB * sizeof(float);
#define FW_PIPE_FWD(TY, GI)                                                                                                             \
    do {                                                                                                                                \
        FW_SET_LDS_ONCE((dwconv_fwd_pipe_kernel<TY, GI>), lds);                                                                         \
        hipLaunchKernelGGL((dwconv_fwd_pipe_kernel<TY, GI>), dim3((unsigned)nb), dim3(256), lds, ST, (const TY*)g1, ld1, w, bias, (TY*)h2, \
                           (TY*)g2, ld2, B, H, W, C, (int)NT);                                                                          \
    } while (0)
        if (dtype == FW_DT_BF16) { if (in_gelu) FW_PIPE_FWD(bf16raw, true); else FW_PIPE_FWD(bf16raw, false); }
        else { if (in_gelu) FW_PIPE_FWD(float, true); else FW_PIPE_FWD(float, false); }
#undef FW_PIPE_FWD
        FW_LAUNCH_RET();
    }
    if (dw_tiled() && H % DT_TY == 0 && (long)B * H * W * C >= dw_tiled_min()) {
        if (dtype == FW_DT_BF16)
            return in_gelu ? dwconv_tile_launch<bf16raw, 0, true>((const bf16raw*)g1, ld1, w, bias, (const bf16raw*)nullptr, (bf16raw*)h2, (bf16raw*)g2, ld2, B, H, W, C, ST)
                           : dwconv_tile_launch<bf16raw, 0>((const bf16raw*)g1, ld1, w, bias, (const bf16raw*)nullptr, (bf16raw*)h2, (bf16raw*)g2, ld2, B, H, W, C, ST);
        return in_gelu ? dwconv_tile_launch<float, 0, true>((const float*)g1, ld1, w, bias, (const float*)nullptr, (float*)h2, (float*)g2, ld2, B, H, W, C, ST)
                       : dwconv_tile_launch<float, 0>((const float*)g1, ld1, w, bias, (const float*)nullptr, (float*)h2, (float*)g2, ld2, B, H, W, C, ST);
    }
    const dim3 grid((unsigned)((grid_for(n) + 7) / 8 * 8));            // multiple of 8: XCD-contiguous block mapping
#define FW_STRIP_FWD(TY, GI) hipLaunchKernelGGL((dwconv_strip_kernel<TY, 0, GI>), grid, dim3(TPB), 0, ST, (const TY*)g1, ld1, w, bias, (const TY*)nullptr, (TY*)h2, (TY*)g2, ld2, B, H, W, C)
    if (dtype == FW_DT_BF16) { if (in_gelu) FW_STRIP_FWD(bf16raw, true); else FW_STRIP_FWD(bf16raw, false); }
    else { if (in_gelu) FW_STRIP_FWD(float, true); else FW_STRIP_FWD(float, false); }
#undef FW_STRIP_FWD
    FW_LAUNCH_RET();
}
extern "C" int fw_dwconv_bwd(int dtype, const void* dh2, long ldg, const void* g1, const void* h1, long ld1, const float* w, void* dh1,
                             long ldo, float* dw, float* dbias, int B, int H, int W, int C, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(dh2 && h1 && w && dh1 && dw && dbias && C % e == 0 && ld1 % e == 0 && ldg % e == 0 && ldo % e == 0 && W % SX == 0);
    FW_CHECK_ARG(ldg == ld1);                      // the data-gradient strip kernel walks dh2 and h1 with one row stride
    const long n = (long)B * H * (W / SX) * (C / 4);
    const int nvg = (C / 4 + WG_VL - 1) / WG_VL;
    const long nst = (long)B * H * (W / SX);
    static const int nstrip_max = getenv("FW_DWWG_NSTRIP") ? atoi(getenv("FW_DWWG_NSTRIP")) : 16;
    int NSTRIP = nstrip_max;                       // strips per thread: fewer on small layers so that enough blocks are in flight
    static const long wg_min_blocks = getenv("FW_DWWG_MIN_BLOCKS") ? atol(getenv("FW_DWWG_MIN_BLOCKS")) : 384;
    while (NSTRIP > 2 && ((nst + (long)NSTRIP * WG_SL - 1) / ((long)NSTRIP * WG_SL)) * nvg < wg_min_blocks) NSTRIP >>= 1;   // every block ends in 1280 atomics: not too many blocks
    const long nsg = (nst + (long)NSTRIP * WG_SL - 1) / ((long)NSTRIP * WG_SL);
    const dim3 gridw((unsigned)(nsg * nvg));
    const bool tiled = dw_tiled() && H % DT_TY == 0 && (long)B * H * W * C >= dw_tiled_min();
    static const int fused = getenv("FW_DWCONV_FUSED_BWD") ? atoi(getenv("FW_DWCONV_FUSED_BWD")) : 1;
    if (tiled && fused && !g1 && (long)H * W * ldg < (1L << 31)) {                    // one pass: data gradient + weight / bias gradient (dwconv_bwd_fused_kernel)
        static const long want = getenv("FW_DWCONV_FUSED_WGS") ? atol(getenv("FW_DWCONV_FUSED_WGS")) : 1024;
        const long ncol = (long)((C + DT_CB - 1) / DT_CB) * ((W + DT_TX - 1) / DT_TX), nrow = (long)B * (H / DT_TY);
        long NT = ncol * nrow / want;               // tiles a workgroup walks: as many as still leave ~`want` workgroups
        NT = NT < 1 ? 1 : (NT > 32 ? 32 : NT);
        long nb = ncol * ((nrow + NT - 1) / NT);
        nb = (nb + 7) / 8 * 8;
        const size_t esz = dtype == FW_DT_BF16 ? 2 : 4;
        const size_t lds = (size_t)(DT_TY + 2) * ((DT_TX + 2) * DT_CB * esz + 128) + (9 * DT_CB + 40 * 256) * sizeof(float);   // halo tile + weights + partials
        if (dtype == FW_DT_BF16) {
            FW_SET_LDS_ONCE((dwconv_bwd_fused_kernel<bf16raw>), lds);
            hipLaunchKernelGGL((dwconv_bwd_fused_kernel<bf16raw>), dim3((unsigned)nb), dim3(256), lds, ST, (const bf16raw*)dh2, ldg, (const bf16raw*)h1, w,
                               (bf16raw*)dh1, ldo, dw, dbias, B, H, W, C, (int)NT);
        } else {
            FW_SET_LDS_ONCE((dwconv_bwd_fused_kernel<float>), lds);
            hipLaunchKernelGGL((dwconv_bwd_fused_kernel<float>), dim3((unsigned)nb), dim3(256), lds, ST, (const float*)dh2, ldg, (const float*)h1, w,
                               (float*)dh1, ldo, dw, dbias, B, H, W, C, (int)NT);
        }
        FW_LAUNCH_RET();
    }
    if (tiled) {
        const int rc = dtype == FW_DT_BF16
            ? dwconv_tile_launch<bf16raw, 1>((const bf16raw*)dh2, ldg, w, (const float*)nullptr, (const bf16raw*)h1, (bf16raw*)dh1, (bf16raw*)nullptr, ldo, B, H, W, C, ST)
            : dwconv_tile_launch<float, 1>((const float*)dh2, ldg, w, (const float*)nullptr, (const float*)h1, (float*)dh1, (float*)nullptr, ldo, B, H, W, C, ST);
        if (rc) return rc;
    }
    // weight gradient: from the activation g1 when the caller kept it, else from the pre-activation h1 (GELU once per element in-kernel)
    if (dtype == FW_DT_BF16) {
        if (!tiled) hipLaunchKernelGGL((dwconv_strip_kernel<bf16raw, 1>), dim3((grid_for(n) + 7) / 8 * 8), dim3(TPB), 0, ST, (const bf16raw*)dh2, ldg, w, (const float*)nullptr,
                           (const bf16raw*)h1, (bf16raw*)dh1, (bf16raw*)nullptr, ldo, B, H, W, C);
        if (g1) hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16raw, false>), gridw, dim3(256), 0, ST, (const bf16raw*)dh2, ldg, (const bf16raw*)g1, ld1, dw, dbias, B, H, W, C, NSTRIP);
        else hipLaunchKernelGGL((dwconv_wgrad_kernel<bf16raw, true>), gridw, dim3(256), 0, ST, (const bf16raw*)dh2, ldg, (const bf16raw*)h1, ld1, dw, dbias, B, H, W, C, NSTRIP);
    } else {
        if (!tiled) hipLaunchKernelGGL((dwconv_strip_kernel<float, 1>), dim3((grid_for(n) + 7) / 8 * 8), dim3(TPB), 0, ST, (const float*)dh2, ldg, w, (const float*)nullptr,
                           (const float*)h1, (float*)dh1, (float*)nullptr, ldo, B, H, W, C);
        if (g1) hipLaunchKernelGGL((dwconv_wgrad_kernel<float, false>), gridw, dim3(256), 0, ST, (const float*)dh2, ldg, (const float*)g1, ld1, dw, dbias, B, H, W, C, NSTRIP);
        else hipLaunchKernelGGL((dwconv_wgrad_kernel<float, true>), gridw, dim3(256), 0, ST, (const float*)dh2, ldg, (const float*)h1, ld1, dw, dbias, B, H, W, C, NSTRIP);
    }
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
    int grid = (int)((rows + rpb - 1) / rpb);
    if (x_dtype != 1 && cols % 4 == 0 && cols <= 512 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 3) == 0) {
        // every block ends in `cols` atomics on the SAME words: 1 024 blocks serialise ~50 us of them on a 28-column matrix -- 256 blocks
        rpb = (int)((rows + 255) / 256);
        if (rpb < 32) rpb = 32;
        grid = (int)((rows + rpb - 1) / rpb);
        hipLaunchKernelGGL(colsum4_kernel, dim3(grid), dim3(TPB), 0, ST, (const float*)x, ldx, out, rows, cols, rpb);
        FW_LAUNCH_RET();
    }
    if (x_dtype == 1) hipLaunchKernelGGL((colsum_t_kernel<bf16raw>), dim3(grid), dim3(TPB), 0, ST, (const bf16raw*)x, ldx, out, rows, cols, rpb);
    else hipLaunchKernelGGL(colsum_kernel, dim3(grid), dim3(TPB), 0, ST, (const float*)x, ldx, out, rows, cols, rpb);
    FW_LAUNCH_RET();
}
extern "C" int fw_inproj_fwd(const float* img, const float* w, const float* bias, float* out, long ldo, int B, int H, int W, int C,
                             float slope, void* stream) {
    FW_CHECK_ARG(img && w && bias && out && C % 4 == 0 && ldo % 4 == 0);
    hipLaunchKernelGGL(inproj_fwd_kernel, dim3(grid_for((long)B * H * W * (C / 4), 2048)), dim3(TPB), (size_t)C * 28 * 4, ST, img, w, bias, out, ldo, B, H, W, C, slope);
    FW_LAUNCH_RET();
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
    hipLaunchKernelGGL(outproj_fwd_kernel, dim3(grid_for((long)B * H * W, 2048)), dim3(TPB), (size_t)C * 27 * 4, ST, fea, ldf, w, bias, img, out, B, H, W, C);
    FW_LAUNCH_RET();
}
extern "C" int fw_outproj_bwd(const float* dout, const float* fea, long ldf, const float* w, float* dfea, long lddf, float* dw,
                              float* db, int B, int H, int W, int C, void* stream) {
    FW_CHECK_ARG(dout && fea && w && dfea && dw && db && C % 4 == 0 && ldf % 4 == 0 && lddf % 4 == 0);
    hipLaunchKernelGGL(outproj_bwd_data_kernel, dim3(grid_for((long)B * H * W * (C / 4), 2048)), dim3(TPB), (size_t)C * 27 * 4, ST, dout, w, dfea, lddf, B, H, W, C);
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
// dst must be pre-initialised when accumulate != 0 (the z range may then be split over blockIdx.y with atomics).
// zstride and the segment [0, max(n, off2 + n2)) rounded up to 4 must lie inside every slab row.
extern "C" int fw_slab_reduce(const float* slab, int nz, long n, long zstride, float* dst, int accumulate, float* dst2, long off2, long n2,
                              void* stream) {
    FW_CHECK_ARG(slab && dst && nz > 0 && n > 0 && zstride % 4 == 0 && ((uintptr_t)slab & 15) == 0);
    FW_CHECK_ARG(!dst2 || (off2 >= n && n2 > 0 && off2 + n2 <= zstride));
    const long end = dst2 ? off2 + n2 : n;
    const long gx = ((end + 3) / 4 + 63) / 64;
    int zper = nz;                                  // plain stores need the whole z range in one block
    if (accumulate) {                               // aim at >= ~1024 blocks, at least 8 slab rows per block
        long splits = 1024 / gx;
        if (splits > nz / 8) splits = nz / 8;
        if (splits < 1) splits = 1;
        zper = (int)((nz + splits - 1) / splits);
    }
    dim3 grid((unsigned)gx, (unsigned)((nz + zper - 1) / zper));
    hipLaunchKernelGGL(slab_reduce_kernel, grid, dim3(256), 0, ST, slab, nz, n, zstride, dst, accumulate, zper, dst2, off2, n2);
    FW_LAUNCH_RET();
}
// One launch for `num` slabs.  tab / prefix are DEVICE arrays laid out as slab_reduce_multi_kernel documents (the host picks zper,
// nchunks = ceil(ceil(max(n, off2 + n2) / 4) / 64) and upw per entry; an entry owns ceil(nchunks * ceil(nz / zper) / (4 * upw)) blocks).
extern "C" int fw_slab_reduce_multi(const void* tab, const void* prefix, int num, long total_blocks, void* stream) {
    FW_CHECK_ARG(tab && prefix && num > 0 && total_blocks > 0 && total_blocks < (1L << 31));
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ST, (const long long*)tab, (const long long*)prefix, num);
    FW_LAUNCH_RET();
}
// One launch for `num` re-layouts; tab / prefix are DEVICE arrays as permute3_multi_kernel documents (entry e owns
// ceil(d0*d1*d2 / 1024) blocks).
extern "C" int fw_permute3_multi(const void* tab, const void* prefix, int num, long total_blocks, void* stream) {
    FW_CHECK_ARG(tab && prefix && num > 0 && total_blocks > 0 && total_blocks < (1L << 31));
    hipLaunchKernelGGL(permute3_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ST, (const long long*)tab, (const long long*)prefix, num);
    FW_LAUNCH_RET();
}
extern "C" int fw_fill(float* p, long n, float v, void* stream) {
    FW_CHECK_ARG(p && n > 0);
    LAUNCH(fill_kernel, n, p, n, v);
}
