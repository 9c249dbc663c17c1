// Window attention of the AirNet hot path (K1 / K2 / K5 of SURVEY.md section 2.2), forward + backward.
//
// Replaces, per (window, head):
//   decoder  WindowAttention.forward      net/decoder_Uformer.py:240-293  (W-MSA + learned frequency selection)
//   encoder  FrequencyWindowAttention     net/encoder_Uformer.py:256-310  (intra / inter band attention)
//   encoder  WindowAttention ("origin")   net/encoder_Uformer.py:152-183
// with the window partition / reverse and the cyclic roll (decoder_Uformer.py:387-409,678-686,721-729)
// folded into the row gather / scatter, the relative-position bias gathered in-kernel from the
// [225, heads] table (:245-251), and the SW-MSA mask (:634-651) rebuilt from the window coordinates
// (-100 where the two tokens lie in different shift regions, exactly as the reference adds it).
//
// MI355X design.  One 64-lane wave owns one (window, head) item end to end; the 64x64 score tile never
// leaves the CU.  Every contraction is an MFMA "NT" product of two k-contiguous LDS tiles
// (fw_common.h), and every product is oriented so that its result -- which the C/D layout delivers
// with 4 consecutive ROWS per lane -- is written back to LDS TRANSPOSED with one 8/16-byte store and
// is then exactly the k-contiguous operand the next product needs:
//     S^T[j][i] = K Q^T        softmax over j = registers + 2 shuffles (columns i live on lanes)
//     O^T[d][i] = V^T P'       V read k-major from its row tile; O^T stored transposed -> [i][d]
// Learned frequency selection: the reference FFTs each 64x64 attention map, masks three radial bands
// and adds lambda_i * band_i back (decoder_Uformer.py:275-288).  Because band0 is the DC bin only and
// the bands partition the plane, band2 = P - mean(P) - band1, so
//     P' = (1 + l2) P - l2/64 + (l1 - l2) B1(P),      B1 = the 0 < |f| <= r/2 disc filter,
// (rows of a softmax sum to 1, hence mean(P) = 1/64) and B1 is a real, self-adjoint operator.  It is
// evaluated as a PARTIAL DFT on MFMA: only |fu| <= 22, 0 <= fv <= 22 are ever formed,
//     T = P Fv (64x64 . 64x32),  X = Fu T (48x64 . 64x32),  Y = Mw * X,
//     Z^T = Y^T Fu^H (32x64 . 64x64),  B1(P)^T = G Z^T (64x32 . 32x64)
// = 176 MFMA 16x16x32 per head-window next to the 64 of QK^T + AV; constant DFT panels are read as
// ready-made fragments from L2.  The backward pass needs B1 twice (for P' in dV and for
// dP = a dP' + c B1(dP')) and gets d(lambda) from three inner products.
#include "fw_common.h"

namespace {

struct AttnArgs {
    const char* q; const char* k; const char* v;      // T, row = token, head h at column h*D
    long ld;                                          // elements, common to q/k/v
    char* out; long ldo;                              // fwd output / bwd: forward output O (unused)
    float* lse;                                       // [items][64]
    const float* bias;                                // [ntab][225][heads]
    const float* coef;                                // [B][heads][3]  (a, b, c) or null
    const char* lfs;                                  // DFT panels (T) followed by mask (float)
    int B, H, W, heads, L, mode, shift;
    float scale;
    int nwin;                                         // B * nWy * nWx
    // backward only
    const char* dout; long lddo;
    char* dq; char* dk; char* dv; long ldd;           // dq/dk/dv rows, head h at column h*D
    char* dk2; char* dv2;                             // second slot for inter (NKT=2) key gradients
    float* dbias;                                     // [ntab][225][heads], same layout as the bias tables
    float* dcoef;                                     // [B][heads][3]
    int chunks;                                       // windows are split into `chunks` per (band, head)
};

// ---- LFS panel offsets (in elements of T), see build_lfs_tables() on the host side -------------
constexpr int NU = 48, NV = 32;
constexpr int OFF_C2 = 0;                    // [32][64]  cos(2pi v j/64)            B-op of T = P Fv
constexpr int OFF_S2N = OFF_C2 + 32 * 64;    // [32][64]  -sin
constexpr int OFF_CU = OFF_S2N + 32 * 64;    // [48][64]  cos(2pi fu i/64)           A-op of X = Fu T
constexpr int OFF_SU = OFF_CU + 48 * 64;     // [48][64]  sin
constexpr int OFF_SUN = OFF_SU + 48 * 64;    // [48][64]  -sin
constexpr int OFF_CH = OFF_SUN + 48 * 64;    // [64][64]  cos(2pi fu i/64) as [i][u] B-op of Z^T = Y^T Fu^H
constexpr int OFF_SH = OFF_CH + 64 * 64;     // [64][64]  sin
constexpr int OFF_SHN = OFF_SH + 64 * 64;    // [64][64]  -sin
constexpr int OFF_GC = OFF_SHN + 64 * 64;    // [64][32]  cos(2pi v j/64) as [j][v]  A-op of out^T = G Z^T
constexpr int OFF_GSN = OFF_GC + 64 * 32;    // [64][32]  -sin
constexpr int OFF_END = OFF_GSN + 64 * 32;   // followed by float mask Mw[48][32]

template <typename T, int D> struct Geo {
    static constexpr int SZ = TT<T>::SZ;
    static constexpr int KB = ((D * SZ + 63) / 64) * 64;     // bytes of K per row, whole 64-byte chunks
    static constexpr int KC = KB / 64;                        // chunks over d
    static constexpr int LDR = KB + 16;                       // row stride of a [64][D] tile
    static constexpr int CB = (D * SZ) % 16 == 0 ? 16 : 8;    // global copy granule
    static constexpr int CH = D * SZ / CB;                    // granules holding data
    static constexpr int SL = KB / CB;                        // granules per padded row
    static constexpr int DT = (D + 15) / 16;                  // 16-row tiles over d
    static constexpr int TILE_D = 64 * LDR;
    static constexpr int JC = 64 * SZ / 64;                   // chunks over 64 tokens
    static constexpr int LDP = 64 * SZ + 16;                  // row stride of a [..][64 tokens] tile
    static constexpr int LDV = 32 * SZ + 16;                  // row stride of a [..][32 v] tile
    static constexpr int VC = 32 * SZ / 64;                   // chunks over 32 v
};

FW_DEV int other_band(int lq, int kt) { return kt < lq ? kt : kt + 1; }   // kt-th band != lq

FW_DEV int kt_slot(int lq, int lk) { return lq < lk ? lq : lq - 1; }        // rank of lq among the bands != lk

// token row (global) of window-local token t for image n, window (wy, wx)
FW_DEV long token_row(int n, int wy, int wx, int t, int H, int W, int shift) {
    int y = wy * 8 + (t >> 3) + shift; if (y >= H) y -= H;
    int x = wx * 8 + (t & 7) + shift;  if (x >= W) x -= W;
    return ((long)n * H + y) * W + x;
}

// Copy the 64 window rows of one head ([64][D]) global -> LDS tile (zero padded to whole chunks).
template <typename T, int D>
FW_DEV void load_tile(char* tile, const char* base, long ld, int n, int wy, int wx, int H, int W, int shift, int col) {
    using G = Geo<T, D>;
    const int l = lane_id();
    for (int idx = l; idx < 64 * G::SL; idx += 64) {
        const int t = idx / G::SL, s = idx % G::SL;
        if (G::CB == 16) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (s < G::CH) v = *reinterpret_cast<const uint4*>(base + (token_row(n, wy, wx, t, H, W, shift) * ld + col) * G::SZ + s * 16);
            *reinterpret_cast<uint4*>(tile + t * G::LDR + s * 16) = v;
        } else {
            uint2 v = make_uint2(0, 0);
            if (s < G::CH) v = *reinterpret_cast<const uint2*>(base + (token_row(n, wy, wx, t, H, W, shift) * ld + col) * G::SZ + s * 8);
            *reinterpret_cast<uint2*>(tile + t * G::LDR + s * 8) = v;
        }
    }
}
// Copy a [64][D] LDS tile -> the 64 window rows of one head.
template <typename T, int D>
FW_DEV void store_tile(const char* tile, char* base, long ld, int n, int wy, int wx, int H, int W, int shift, int col) {
    using G = Geo<T, D>;
    const int l = lane_id();
    for (int idx = l; idx < 64 * G::CH; idx += 64) {
        const int t = idx / G::CH, s = idx % G::CH;
        char* dst = base + (token_row(n, wy, wx, t, H, W, shift) * ld + col) * G::SZ;
        if (G::CB == 16) *reinterpret_cast<uint4*>(dst + s * 16) = *reinterpret_cast<const uint4*>(tile + t * G::LDR + s * 16);
        else *reinterpret_cast<uint2*>(dst + s * 8) = *reinterpret_cast<const uint2*>(tile + t * G::LDR + s * 8);
    }
}

// bias + shift mask for score element (query i, key j) of window (wy, wx)
FW_DEV float bias_mask(const float* tab, int heads, int h, int i, int j, int shift, bool last_y, bool last_x) {
    const int dy = (i >> 3) - (j >> 3) + 7, dx = (i & 7) - (j & 7) + 7;
    float b = tab[(dy * 15 + dx) * heads + h];
    if (shift > 0) {
        const int s = 8 - shift;
        const int ri = (last_y ? ((i >> 3) < s ? 1 : 2) : 0) * 3 + (last_x ? ((i & 7) < s ? 1 : 2) : 0);
        const int rj = (last_y ? ((j >> 3) < s ? 1 : 2) : 0) * 3 + (last_x ? ((j & 7) < s ? 1 : 2) : 0);
        if (ri != rj) b += -100.0f;
    }
    return b;
}

// sum / max over the rows (j) of a [16*JT][64] score block held as acc[jt][it]: registers + 2 shuffles
FW_DEV float col_reduce_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
FW_DEV float col_reduce_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// ---- B1: the disc band filter on a 64x64 tile held transposed in registers (in[jt][it] = A^T[j][i]) ----
// scrA / scrB: two LDS scratch regions of >= 64*LDP and >= 2*64*LDV bytes.  out[jt][it] = B1(A)^T[j][i].
template <typename T>
FW_DEV void band_filter(const f32x4 (&in)[4][4], f32x4 (&out)[4][4], char* scrA, char* scrB, const char* lfs) {
    using G = Geo<T, 56>;
    constexpr int SZ = G::SZ, LDP = G::LDP, LDV = G::LDV, JC = G::JC, VC = G::VC;
    const char* tab = lfs;
    const float* Mw = reinterpret_cast<const float*>(lfs + (size_t)OFF_END * SZ);
    const int l = lane_id();
    // Ps[i][j] <- in^T   (scrA)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) store_acc_T<T>(scrA, LDP, jt * 16, it * 16, in[jt][it]);
    __syncthreads();
    // T[i][v] = sum_j P[i][j] Fv[j][v]   (Tr with cos, Ti with -sin); store transposed -> Ts[v][i] (scrB)
    {
        f32x4 tr[4][2], ti[4][2];
        zero_acc(tr); zero_acc(ti);
#pragma unroll 1
        for (int c = 0; c < JC; ++c) {
            uint4 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = frag_kc(scrA, LDP, m * 16, c);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint4 bc = frag_kc(tab + (size_t)OFF_C2 * SZ, 64 * SZ, n * 16, c);
                const uint4 bs = frag_kc(tab + (size_t)OFF_S2N * SZ, 64 * SZ, n * 16, c);
#pragma unroll
                for (int m = 0; m < 4; ++m) { mma_chunk<T>(tr[m][n], a[m], bc); mma_chunk<T>(ti[m][n], a[m], bs); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                store_acc_T<T>(scrB, LDP, m * 16, n * 16, tr[m][n]);                 // Tr^T [v][i]
                store_acc_T<T>(scrB + 32 * LDP, LDP, m * 16, n * 16, ti[m][n]);      // Ti^T [v][i]
            }
    }
    __syncthreads();
    // X[u][v] = sum_i Fu[u][i] T[i][v]:  Xr = Cu Tr + Su Ti,  Xi = Cu Ti - Su Tr;  Y = Mw * X -> Ys[v][u] (scrA)
    {
        f32x4 xr[3][2], xi[3][2];
        zero_acc(xr); zero_acc(xi);
#pragma unroll 1
        for (int c = 0; c < JC; ++c) {
            uint4 br[2], bi[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) { br[n] = frag_kc(scrB, LDP, n * 16, c); bi[n] = frag_kc(scrB + 32 * LDP, LDP, n * 16, c); }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const uint4 ac = frag_kc(tab + (size_t)OFF_CU * SZ, 64 * SZ, m * 16, c);
                const uint4 as = frag_kc(tab + (size_t)OFF_SU * SZ, 64 * SZ, m * 16, c);
                const uint4 an = frag_kc(tab + (size_t)OFF_SUN * SZ, 64 * SZ, m * 16, c);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    mma_chunk<T>(xr[m][n], ac, br[n]); mma_chunk<T>(xr[m][n], as, bi[n]);
                    mma_chunk<T>(xi[m][n], ac, bi[n]); mma_chunk<T>(xi[m][n], an, br[n]);
                }
            }
        }
        __syncthreads();
        // zero the k padding u = 48..63 of Ys (two [32][64] panels)
        for (int idx = l; idx < 64 * 4; idx += 64) {
            const int row = idx >> 2, part = idx & 3;          // 64 rows (2 panels x 32), 16 pad elements in 4 parts
            char* p = scrA + row * LDP + 48 * SZ + part * 4 * SZ;
            if (SZ == 4) *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0); else *reinterpret_cast<uint2*>(p) = make_uint2(0, 0);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = Mw[(m * 16 + ((l >> 4) << 2) + r) * NV + n * 16 + (l & 15)];
                store_acc_T<T>(scrA, LDP, m * 16, n * 16, xr[m][n] * w);             // Yr^T [v][u]
                store_acc_T<T>(scrA + 32 * LDP, LDP, m * 16, n * 16, xi[m][n] * w);  // Yi^T [v][u]
            }
    }
    __syncthreads();
    // Z^T[v][i] = sum_u Y^T[v][u] FuH[i][u]:  Zr = Yr C - Yi S,  Zi = Yi C + Yr S;  store transposed -> Zs[i][v] (scrB)
    {
        f32x4 zr[2][4], zi[2][4];
        zero_acc(zr); zero_acc(zi);
#pragma unroll 1
        for (int c = 0; c < JC; ++c) {
            uint4 ar[2], ai[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) { ar[m] = frag_kc(scrA, LDP, m * 16, c); ai[m] = frag_kc(scrA + 32 * LDP, LDP, m * 16, c); }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const uint4 bc = frag_kc(tab + (size_t)OFF_CH * SZ, 64 * SZ, n * 16, c);
                const uint4 bs = frag_kc(tab + (size_t)OFF_SH * SZ, 64 * SZ, n * 16, c);
                const uint4 bn = frag_kc(tab + (size_t)OFF_SHN * SZ, 64 * SZ, n * 16, c);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    mma_chunk<T>(zr[m][n], ar[m], bc); mma_chunk<T>(zr[m][n], ai[m], bn);
                    mma_chunk<T>(zi[m][n], ai[m], bc); mma_chunk<T>(zi[m][n], ar[m], bs);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                store_acc_T<T>(scrB, LDV, m * 16, n * 16, zr[m][n]);                 // Zr [i][v]
                store_acc_T<T>(scrB + 64 * LDV, LDV, m * 16, n * 16, zi[m][n]);      // Zi [i][v]
            }
    }
    __syncthreads();
    // out^T[j][i] = sum_v Gc[j][v] Zr[i][v] + Gsn[j][v] Zi[i][v]
    zero_acc(out);
#pragma unroll 1
    for (int c = 0; c < VC; ++c) {
        uint4 br[4], bi[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { br[n] = frag_kc(scrB, LDV, n * 16, c); bi[n] = frag_kc(scrB + 64 * LDV, LDV, n * 16, c); }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint4 ac = frag_kc(tab + (size_t)OFF_GC * SZ, 32 * SZ, m * 16, c);
            const uint4 as = frag_kc(tab + (size_t)OFF_GSN * SZ, 32 * SZ, m * 16, c);
#pragma unroll
            for (int n = 0; n < 4; ++n) { mma_chunk<T>(out[m][n], ac, br[n]); mma_chunk<T>(out[m][n], as, bi[n]); }
        }
    }
    __syncthreads();
}

template <typename T, int D, int NKT, int LFS> struct Smem {
    using G = Geo<T, D>;
    static constexpr int LDPK = NKT * 64 * G::SZ + 16;                       // P' tile [64 i][NKT*64 j]
    static constexpr int SCR = LFS == 2 ? (2 * 64 * G::LDV > 64 * G::LDP ? 2 * 64 * G::LDV : 64 * G::LDP) : 0;
    static constexpr int RA = (64 * LDPK > SCR ? 64 * LDPK : SCR) > G::TILE_D ? (64 * LDPK > SCR ? 64 * LDPK : SCR) : G::TILE_D;
    static constexpr int RB = (SCR > G::TILE_D ? SCR : G::TILE_D);
    static constexpr int FWD_BYTES = RA + RB * NKT + G::TILE_D * NKT;        // A | B (K tiles) | V tiles
};

// =====================================================================================================
// forward
// =====================================================================================================
template <typename T, int D, int NKT, int LFS>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = Smem<T, D, NKT, LFS>;
    constexpr int SZ = G::SZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* rA = smem;                        // Q -> P' (and LFS scratch A)
    char* rB = smem + S::RA;                // K tiles (kt) -> LFS scratch B -> O staging
    char* rV = rB + S::RB * NKT;            // V tiles
    const int l = lane_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    const int items = a.nwin * a.L * a.heads;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int h = item % a.heads;
        const int lq = (item / a.heads) % a.L;
        const int win = item / (a.heads * a.L);
        const int b = win / nW, w = win % nW, wy = w / nWx, wx = w % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        load_tile<T, D>(rA, a.q, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int lk = a.mode == 0 ? lq : other_band(lq, kt);
            load_tile<T, D>(rB + kt * S::RB, a.k, a.ld, lk * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
            load_tile<T, D>(rV + kt * G::TILE_D, a.v, a.ld, lk * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
        }
        __syncthreads();
        // S^T[j][i] = sum_d K[j][d] Q[i][d]
        f32x4 p[4 * NKT][4];
        zero_acc(p);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            for (int c = 0; c < G::KC; ++c) {
                uint4 ak[4], bq[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) ak[m] = frag_kc(rB + kt * S::RB, G::LDR, m * 16, c);
#pragma unroll
                for (int n = 0; n < 4; ++n) bq[n] = frag_kc(rA, G::LDR, n * 16, c);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma_chunk<T>(p[4 * kt + m][n], ak[m], bq[n]);
            }
        // scale, bias, mask, softmax over j
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = it * 16 + (l & 15);
            float mx = -3.0e38f;
#pragma unroll
            for (int jt = 0; jt < 4 * NKT; ++jt) {
                const int kt = jt >> 2;
                const int lk = a.mode == 0 ? lq : other_band(lq, kt);
                const float* tab = a.bias + (size_t)(lq * a.L + lk) * 225 * a.heads;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = (jt & 3) * 16 + ((l >> 4) << 2) + r;
                    const float s = p[jt][it][r] * a.scale + bias_mask(tab, a.heads, h, i, j, a.shift, last_y, last_x);
                    p[jt][it][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
            mx = col_reduce_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4 * NKT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __expf(p[jt][it][r] - mx); p[jt][it][r] = e; sum += e; }
            sum = col_reduce_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int jt = 0; jt < 4 * NKT; ++jt) p[jt][it] *= inv;
            if ((l >> 4) == 0) a.lse[(size_t)item * 64 + i] = mx + __logf(sum);
        }
        __syncthreads();                    // Q / K tiles are dead from here on
        if constexpr (LFS >= 1) {
            static_assert(NKT == 1, "frequency selection acts on one 64x64 map");
            const float* cf = a.coef + ((size_t)b * a.heads + h) * 3;
            const float ca = cf[0], cb = cf[1], cc = cf[2];
            if constexpr (LFS == 2) {
                f32x4 f1[4][4];
                band_filter<T>(p, f1, rA, rB, a.lfs);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int it = 0; it < 4; ++it) p[jt][it] = p[jt][it] * ca + cb + f1[jt][it] * cc;
            } else {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int it = 0; it < 4; ++it) p[jt][it] = p[jt][it] * ca + cb;
            }
        }
        // P'[i][j] -> rA (transposed store), then O^T[d][i] = sum_j V[j][d] P'[i][j]
#pragma unroll
        for (int jt = 0; jt < 4 * NKT; ++jt)
#pragma unroll
            for (int it = 0; it < 4; ++it) store_acc_T<T>(rA, S::LDPK, jt * 16, it * 16, p[jt][it]);
        __syncthreads();
        f32x4 o[G::DT][4];
        zero_acc(o);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            for (int c = 0; c < G::JC; ++c) {
                uint4 av[G::DT], bp[4];
#pragma unroll
                for (int m = 0; m < G::DT; ++m) av[m] = frag_km<T>(rV + kt * G::TILE_D, G::LDR, m * 16, c);
#pragma unroll
                for (int n = 0; n < 4; ++n) bp[n] = frag_kc(rA, S::LDPK, n * 16, kt * G::JC + c);
#pragma unroll
                for (int m = 0; m < G::DT; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma_chunk<T>(o[m][n], av[m], bp[n]);
            }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < G::DT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) store_acc_T<T>(rB, G::LDR, m * 16, n * 16, o[m][n]);      // O [i][d]
        __syncthreads();
        store_tile<T, D>(rB, a.out, a.ldo, nq, wy, wx, a.H, a.W, a.shift, h * D);
        __syncthreads();
    }
}

// =====================================================================================================
// backward
// =====================================================================================================
template <typename T, int D, int NKT, int LFS> struct SmemB {
    using G = Geo<T, D>;
    static constexpr int SCR = 2 * 64 * G::LDV > 64 * G::LDP ? 2 * 64 * G::LDV : 64 * G::LDP;   // >= one [64][64] tile
    static constexpr int OFF_Q = 0, OFF_DO = G::TILE_D, OFF_K = 2 * G::TILE_D, OFF_V = 3 * G::TILE_D;
    static constexpr int OFF_X = 4 * G::TILE_D, OFF_Y = OFF_X + SCR;
    static constexpr int OFF_DB = OFF_Y + SCR;                                                 // float [NKT][64][64]
    static constexpr int OFF_DI = OFF_DB + NKT * 64 * 64 * 4;                                  // float [64]
    static constexpr int OFF_BIN = OFF_DI + 256;                                               // float [225] (+pad)
    static constexpr int BYTES = OFF_BIN + 1024;
};

template <typename T, int D, int NKT, int LFS>
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = SmemB<T, D, NKT, LFS>;
    constexpr int SZ = G::SZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem + S::OFF_Q; char* sDO = smem + S::OFF_DO; char* sK = smem + S::OFF_K; char* sV = smem + S::OFF_V;
    char* sX = smem + S::OFF_X; char* sY = smem + S::OFF_Y;
    float* sDB = reinterpret_cast<float*>(smem + S::OFF_DB);
    float* sDI = reinterpret_cast<float*>(smem + S::OFF_DI);
    const int l = lane_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    // grid: x = chunk, y = head, z = query band
    const int h = blockIdx.y, lq = blockIdx.z;
    for (int idx = l; idx < NKT * 4096; idx += 64) sDB[idx] = 0.f;
    __syncthreads();
    for (int win = blockIdx.x; win < a.nwin; win += gridDim.x) {
        const int b = win / nW, w = win % nW, wy = w / nWx, wx = w % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        const size_t item = ((size_t)win * a.L + lq) * a.heads + h;
        load_tile<T, D>(sQ, a.q, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
        load_tile<T, D>(sDO, a.dout, a.lddo, nq, wy, wx, a.H, a.W, a.shift, h * D);
        float lse[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) lse[it] = a.lse[item * 64 + it * 16 + (l & 15)];
        float di[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (NKT > 1) {
            // D_i = sum_d O[i][d] dO[i][d]   (rowsum(P o dP) == rowsum(O o dO) when P' = P)
            const long row = token_row(nq, wy, wx, l, a.H, a.W, a.shift);
            const T* op = reinterpret_cast<const T*>(a.out) + row * a.ldo + h * D;
            const T* gp = reinterpret_cast<const T*>(a.dout) + row * a.lddo + h * D;
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += TT<T>::ld(op + d) * TT<T>::ld(gp + d);
            sDI[l] = s;
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) di[it] = sDI[it * 16 + (l & 15)];
        }
        f32x4 dq[G::DT][4];
        zero_acc(dq);
#pragma unroll 1
        for (int kt = 0; kt < NKT; ++kt) {
            const int lk = a.mode == 0 ? lq : other_band(lq, kt);
            const int nk = lk * a.B + b;
            const int tabid = lq * a.L + lk;
            __syncthreads();
            load_tile<T, D>(sK, a.k, a.ld, nk, wy, wx, a.H, a.W, a.shift, h * D);
            load_tile<T, D>(sV, a.v, a.ld, nk, wy, wx, a.H, a.W, a.shift, h * D);
            __syncthreads();
            // P^T[j][i] = exp(scale * K Q^T + bias + mask - lse_i)
            f32x4 p[4][4], dp[4][4];
            const float* tab = a.bias + (size_t)tabid * 225 * a.heads;
            auto compute_p = [&]() {
                asm volatile("" ::: "memory");            // do not keep the 64 gathered bias values live across the band filter
                zero_acc(p);
                mma_tiles<T, 4, 4>(p, sK, G::LDR, 0, sQ, G::LDR, 0, G::KC);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int i = it * 16 + (l & 15);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = jt * 16 + ((l >> 4) << 2) + r;
                            p[jt][it][r] = __expf(p[jt][it][r] * a.scale + bias_mask(tab, a.heads, h, i, j, a.shift, last_y, last_x) - lse[it]);
                        }
                }
            };
            compute_p();
            float ca = 1.f, cc = 0.f;
            if constexpr (LFS >= 1) {
                const float* cf = a.coef + ((size_t)b * a.heads + h) * 3;
                ca = cf[0]; const float cb = cf[1]; cc = cf[2];
                // P'^T = a P^T + b (+ c B1(P)^T)  -> sX as [j][i] (k = i contiguous)
                if constexpr (LFS == 2) {
                    f32x4 f[4][4];
                    band_filter<T>(p, f, sX, sY, a.lfs);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int it = 0; it < 4; ++it) store_acc_N<T>(sX, G::LDP, jt * 16, it * 16, p[jt][it] * ca + cb + f[jt][it] * cc);
                } else {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int it = 0; it < 4; ++it) store_acc_N<T>(sX, G::LDP, jt * 16, it * 16, p[jt][it] * ca + cb);
                }
            } else {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int it = 0; it < 4; ++it) store_acc_N<T>(sX, G::LDP, jt * 16, it * 16, p[jt][it]);
            }
            __syncthreads();
            // dV^T[d][j] = sum_i dO[i][d] P'[i][j]   (A = dO read k-major, B = P'^T rows j)
            {
                f32x4 dv[G::DT][4];
                zero_acc(dv);
                for (int c = 0; c < G::JC; ++c) {
                    uint4 am[G::DT], bn[4];
#pragma unroll
                    for (int m = 0; m < G::DT; ++m) am[m] = frag_km<T>(sDO, G::LDR, m * 16, c);
#pragma unroll
                    for (int n = 0; n < 4; ++n) bn[n] = frag_kc(sX, G::LDP, n * 16, c);
#pragma unroll
                    for (int m = 0; m < G::DT; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) mma_chunk<T>(dv[m][n], am[m], bn[n]);
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < G::DT; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) store_acc_T<T>(sY, G::LDR, m * 16, n * 16, dv[m][n]);   // dV [j][d]
                __syncthreads();
                char* dvp = (NKT > 1 && kt_slot(lq, lk) == 1) ? a.dv2 : a.dv;
                store_tile<T, D>(sY, dvp, a.ldd, nk, wy, wx, a.H, a.W, a.shift, h * D);
                __syncthreads();
            }
            // dP'^T[j][i] = sum_d V[j][d] dO[i][d]   (after dV: P' and its scratch are dead, keeps the live set small)
            zero_acc(dp);
            mma_tiles<T, 4, 4>(dp, sV, G::LDR, 0, sDO, G::LDR, 0, G::KC);
            if constexpr (LFS >= 1) {
                // G^T = B1(dP')^T ; d(a,b,c) = (<dP',P>, sum dP', <G,P>) ; dP = a dP' + c G
                float s1 = 0.f, s2 = 0.f, s3 = 0.f;
                if constexpr (LFS == 2) {
                    f32x4 g[4][4];
                    band_filter<T>(dp, g, sX, sY, a.lfs);
                    compute_p();                              // P is cheaper to rebuild (32 MFMA) than to keep live across the filter
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int it = 0; it < 4; ++it)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                s1 += dp[jt][it][r] * p[jt][it][r];
                                s2 += dp[jt][it][r];
                                s3 += g[jt][it][r] * p[jt][it][r];
                                dp[jt][it][r] = dp[jt][it][r] * ca + g[jt][it][r] * cc;
                            }
                } else {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int it = 0; it < 4; ++it)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                s1 += dp[jt][it][r] * p[jt][it][r];
                                s2 += dp[jt][it][r];
                                dp[jt][it][r] = dp[jt][it][r] * ca;
                            }
                }
                s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
                if (l == 0) {
                    float* dc = a.dcoef + ((size_t)b * a.heads + h) * 3;
                    atomicAdd(dc, s1); atomicAdd(dc + 1, s2); atomicAdd(dc + 2, s3);
                }
            }
            // D_i (NKT == 1: straight from registers), dS^T = P^T o (dP^T - D_i)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if constexpr (NKT == 1) {
                    float s = 0.f;
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s += p[jt][it][r] * dp[jt][it][r];
                    di[it] = col_reduce_sum(s);
                }
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ds = p[jt][it][r] * (dp[jt][it][r] - di[it]);
                        dp[jt][it][r] = ds;
                        sDB[kt * 4096 + (it * 16 + (l & 15)) * 64 + jt * 16 + ((l >> 4) << 2) + r] += ds;
                    }
            }
            // dS -> sX as [i][j] (transposed store, k = j) and sY as [j][i] (plain store, k = i)
            __syncthreads();
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    store_acc_T<T>(sX, G::LDP, jt * 16, it * 16, dp[jt][it]);
                    store_acc_N<T>(sY, G::LDP, jt * 16, it * 16, dp[jt][it]);
                }
            __syncthreads();
            // dQ^T[d][i] += sum_j K[j][d] dS[i][j] ;  dK^T[d][j] = sum_i Q[i][d] dS[i][j]
            f32x4 dk[G::DT][4], dql[G::DT][4];
            zero_acc(dk);
            zero_acc(dql);
            for (int c = 0; c < G::JC; ++c) {
                uint4 ak[G::DT], aq[G::DT], bx[4], by[4];
#pragma unroll
                for (int m = 0; m < G::DT; ++m) { ak[m] = frag_km<T>(sK, G::LDR, m * 16, c); aq[m] = frag_km<T>(sQ, G::LDR, m * 16, c); }
#pragma unroll
                for (int n = 0; n < 4; ++n) { bx[n] = frag_kc(sX, G::LDP, n * 16, c); by[n] = frag_kc(sY, G::LDP, n * 16, c); }
#pragma unroll
                for (int m = 0; m < G::DT; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) { mma_chunk<T>(dql[m][n], ak[m], bx[n]); mma_chunk<T>(dk[m][n], aq[m], by[n]); }
            }
#pragma unroll
            for (int m = 0; m < G::DT; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) dq[m][n] += dql[m][n];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < G::DT; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) store_acc_T<T>(sX, G::LDR, m * 16, n * 16, dk[m][n] * a.scale);   // dK [j][d]
            __syncthreads();
            char* dkp = (NKT > 1 && kt_slot(lq, lk) == 1) ? a.dk2 : a.dk;
            store_tile<T, D>(sX, dkp, a.ldd, nk, wy, wx, a.H, a.W, a.shift, h * D);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < G::DT; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) store_acc_T<T>(sY, G::LDR, m * 16, n * 16, dq[m][n] * a.scale);         // dQ [i][d]
        __syncthreads();
        store_tile<T, D>(sY, a.dq, a.ldd, nq, wy, wx, a.H, a.W, a.shift, h * D);
        __syncthreads();
    }
    // flush the bias-gradient accumulators: fold the 64x64 pairs into the 225 relative positions on chip, then one
    // atomic per bin into the parameter layout [table][225][heads] (all workgroups of a head hit the same 225 words)
    float* bins = reinterpret_cast<float*>(smem + S::OFF_BIN);
    for (int kt = 0; kt < NKT; ++kt) {
        __syncthreads();
        for (int idx = l; idx < 225; idx += 64) bins[idx] = 0.f;
        __syncthreads();
        for (int idx = l; idx < 4096; idx += 64) {
            const int i = idx >> 6, j = idx & 63;
            atomicAdd(&bins[((i >> 3) - (j >> 3) + 7) * 15 + (i & 7) - (j & 7) + 7], sDB[kt * 4096 + idx]);
        }
        __syncthreads();
        const int lk = a.mode == 0 ? lq : other_band(lq, kt);
        float* dst = a.dbias + (size_t)(lq * a.L + lk) * 225 * a.heads;
        for (int idx = l; idx < 225; idx += 64) atomicAdd(dst + idx * a.heads + h, bins[idx]);
    }
}


template <typename T, int D, int NKT, int LFS>
int fwd_launch(const AttnArgs& a, hipStream_t st) {
    using S = Smem<T, D, NKT, LFS>;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<T, D, NKT, LFS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, S::FWD_BYTES);
        done = true;
    }
    const int items = a.nwin * a.L * a.heads;
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, NKT, LFS>), dim3(items < 8192 ? items : 8192), dim3(64), S::FWD_BYTES, st, a);
    FW_LAUNCH_RET();
}
template <typename T, int D, int NKT, int LFS>
int bwd_launch(const AttnArgs& a, hipStream_t st) {
    using S = SmemB<T, D, NKT, LFS>;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<T, D, NKT, LFS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, S::BYTES);
        done = true;
    }
    hipLaunchKernelGGL((attn_bwd_kernel<T, D, NKT, LFS>), dim3(a.chunks, a.heads, a.L), dim3(64), S::BYTES, st, a);
    FW_LAUNCH_RET();
}

template <typename T>
int dispatch(bool bwd, int D, int nkt, int lfs, const AttnArgs& a, hipStream_t st) {
#define FW_ATT(DD, KK, FF)                                                                     \
    if (D == DD && nkt == KK && lfs == FF)                                                     \
        return bwd ? bwd_launch<T, DD, KK, FF>(a, st) : fwd_launch<T, DD, KK, FF>(a, st);
    FW_ATT(56, 1, 0) FW_ATT(56, 1, 1) FW_ATT(56, 1, 2) FW_ATT(28, 1, 0) FW_ATT(28, 2, 0)
#undef FW_ATT
    return -1000;      // unsupported (head_dim, key tiles, lfs) combination
}

}  // namespace

extern "C" int fw_attn_lfs_table_elems() { return OFF_END; }

// dtype: 0 f32 / 1 bf16.  head_dim D in {56, 28}.  nkt: key tiles of 64 (2 = inter-band, L = 3).
// lfs: 0 none, 1 affine (all_DC / all_2_bands), 2 affine + disc filter (all_3_bands).
extern "C" int fw_attn_fwd(int dtype, int D, int nkt, int lfs, const void* q, const void* k, const void* v, long ld,
                           void* out, long ldo, float* lse, const float* bias, const float* coef, const void* lfs_tab,
                           int B, int H, int W, int heads, int L, int mode, int shift, float scale, void* stream) {
    FW_CHECK_ARG(q && k && v && out && lse && bias);
    FW_CHECK_ARG(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8 && B > 0 && heads > 0 && L >= 1 && L <= 3);
    FW_CHECK_ARG(shift >= 0 && shift < 8 && (shift == 0 || (H > 8 && W > 8)));
    FW_CHECK_ARG(lfs == 0 || (coef && nkt == 1 && L == 1));
    FW_CHECK_ARG(lfs != 2 || lfs_tab);
    FW_CHECK_ARG((mode == 0 && nkt == 1) || (mode == 1 && nkt == L - 1 && L >= 2));
    const int sz = dtype == FW_DT_BF16 ? 2 : 4;
    FW_CHECK_ARG((ld * sz) % 8 == 0 && (ldo * sz) % 8 == 0);
    AttnArgs a = {};
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld = ld; a.out = (char*)out; a.ldo = ldo;
    a.lse = lse; a.bias = bias; a.coef = coef; a.lfs = (const char*)lfs_tab;
    a.B = B; a.H = H; a.W = W; a.heads = heads; a.L = L; a.mode = mode; a.shift = shift; a.scale = scale;
    a.nwin = B * (H / 8) * (W / 8);
    hipStream_t st = (hipStream_t)stream;
    return dtype == FW_DT_BF16 ? dispatch<bf16raw>(false, D, nkt, lfs, a, st) : dispatch<float>(false, D, nkt, lfs, a, st);
}

// Backward.  dq/dk/dv: rows = tokens, head h at column h*D, row stride ldd (one [T][3C] buffer works:
// pass dq, dq + C, dq + 2C).  dk2/dv2: second slot for inter-band key gradients (nkt == 2), summed by
// the caller.  dbias: f32 [L*L][225][heads] (the tables' layout), dcoef: [B][heads][3] f32 -- both ACCUMULATED into.
extern "C" int fw_attn_bwd(int dtype, int D, int nkt, int lfs, const void* q, const void* k, const void* v, long ld,
                           const void* out, long ldo, const void* dout, long lddo, const float* lse, const float* bias,
                           const float* coef, const void* lfs_tab, void* dq, void* dk, void* dv, void* dk2, void* dv2,
                           long ldd, float* dbias, float* dcoef, int B, int H, int W, int heads, int L, int mode,
                           int shift, float scale, void* stream) {
    FW_CHECK_ARG(q && k && v && dout && lse && bias && dq && dk && dv && dbias);
    FW_CHECK_ARG(H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8 && B > 0 && heads > 0 && L >= 1 && L <= 3);
    FW_CHECK_ARG(shift >= 0 && shift < 8 && (shift == 0 || (H > 8 && W > 8)));
    FW_CHECK_ARG(lfs == 0 || (coef && dcoef && nkt == 1 && L == 1));
    FW_CHECK_ARG(lfs != 2 || lfs_tab);
    FW_CHECK_ARG((mode == 0 && nkt == 1) || (mode == 1 && nkt == L - 1 && L >= 2));
    FW_CHECK_ARG(nkt == 1 || (out && dk2 && dv2));
    const int sz = dtype == FW_DT_BF16 ? 2 : 4;
    FW_CHECK_ARG((ld * sz) % 8 == 0 && (lddo * sz) % 8 == 0 && (ldd * sz) % 8 == 0);
    AttnArgs a = {};
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld = ld; a.out = (char*)out; a.ldo = ldo;
    a.lse = (float*)lse; a.bias = bias; a.coef = coef; a.lfs = (const char*)lfs_tab;
    a.B = B; a.H = H; a.W = W; a.heads = heads; a.L = L; a.mode = mode; a.shift = shift; a.scale = scale;
    a.nwin = B * (H / 8) * (W / 8);
    a.dout = (const char*)dout; a.lddo = lddo; a.dq = (char*)dq; a.dk = (char*)dk; a.dv = (char*)dv;
    a.dk2 = (char*)dk2; a.dv2 = (char*)dv2; a.ldd = ldd; a.dbias = dbias; a.dcoef = dcoef;
    int chunks = 2048 / (heads * L);
    if (chunks < 1) chunks = 1;
    if (chunks > a.nwin) chunks = a.nwin;
    a.chunks = chunks;
    hipStream_t st = (hipStream_t)stream;
    return dtype == FW_DT_BF16 ? dispatch<bf16raw>(true, D, nkt, lfs, a, st) : dispatch<float>(true, D, nkt, lfs, a, st);
}
