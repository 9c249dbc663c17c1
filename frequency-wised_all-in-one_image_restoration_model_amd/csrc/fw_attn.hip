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
// MI355X design.  One workgroup of 4 waves owns one (window, head) item end to end; the 64x64 score tile
// never leaves the CU.  Wave w owns the 16-column strip w of every 64-wide product (queries 16w..16w+15 of
// S^T / P / dP / dS, keys 16w..16w+15 of dV / dK), so softmax, D_i and the bias-gradient accumulators are
// wave-local and the register tile is 4 MFMA tiles instead of 16; the A-operands (K, V, dO, Q, DFT panels) are
// shared through LDS.  2 workgroups fit a CU (bf16): 8 waves hide each other's LDS / L2 latency.
// Every contraction is an MFMA "NT" product of two k-contiguous LDS tiles (fw_common.h), and every product is
// oriented so that its result -- which the C/D layout delivers with 4 consecutive ROWS per lane -- is written
// back to LDS TRANSPOSED with one 8/16-byte store and is then exactly the k-contiguous operand the next product
// needs; operands needed along the other axis are read with the transposing LDS read (ds_read_b64_tr_b16):
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
#include <stdlib.h>

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
    int dq_pad;                                       // zero columns written behind the LAST head's dq columns (0, or up to the next multiple of 8)
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

constexpr int NWV = 4;                       // waves per workgroup = 16-wide strips of the 64x64 tile
constexpr int NTH = NWV * 64;

FW_DEV int wave_id() { return threadIdx.x >> 6; }

// LDS traffic that stays inside one wave (a strip written and re-read by its owner) needs no block barrier: LDS
// instructions of a wave execute in order, the fence keeps the compiler from moving accesses across and drains the queue.
FW_DEV void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for LDS traffic ONLY.  `__syncthreads()` also drains the vector-memory queue (s_waitcnt vmcnt(0)): it would
// wait for the tiles being prefetched for the NEXT window and for the acknowledgements of the stores just issued -- three exposed
// global round trips per window in the v2 kernels.  LDS consistency needs the DS queue drained and the barrier, nothing more.
FW_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Copy the 64 window rows of one head ([64][D]) global -> LDS tile (zero padded to whole chunks); whole workgroup.
// Two phases: issue() only loads (clamped coordinates, no lane-varying branch, nothing consumes the data), commit() zero-pads
// and writes LDS -- so the loads of SEVERAL tiles (Q, K, V, dO) are in flight together instead of one memory latency each.
template <typename T, int D> struct TileLoad {
    using G = Geo<T, D>;
    static constexpr int NI = (64 * G::SL + NTH - 1) / NTH;
    uint4 r[NI];
    FW_MEM void issue(const char* base, long ld, int n, int wy, int wx, int H, int W, int shift, int col) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = threadIdx.x + i * NTH;
            const int t = (idx / G::SL) & 63, s = idx % G::SL;
            const char* p = base + (token_row(n, wy, wx, t, H, W, shift) * ld + col) * G::SZ + (s < G::CH ? s : 0) * G::CB;
            if (G::CB == 16) r[i] = *reinterpret_cast<const uint4*>(p);
            else { const uint2 v = *reinterpret_cast<const uint2*>(p); r[i].x = v.x; r[i].y = v.y; }
        }
    }
    FW_MEM void commit(char* tile) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = threadIdx.x + i * NTH;
            if (idx >= 64 * G::SL) continue;
            const int t = idx / G::SL, s = idx % G::SL;
            const bool ok = s < G::CH;
            if (G::CB == 16) *reinterpret_cast<uint4*>(tile + t * G::LDR + s * 16) = ok ? r[i] : make_uint4(0, 0, 0, 0);
            else *reinterpret_cast<uint2*>(tile + t * G::LDR + s * 8) = ok ? make_uint2(r[i].x, r[i].y) : make_uint2(0, 0);
        }
    }
};
// Copy rows r0..r0+15 of a [64][D] LDS tile -> the matching window rows of one head; ONE wave (its own strip).
// extra: granules of the tile's zero padding (columns D .. D + extra * CB / SZ - 1: products of zero-padded operand columns) copied too
template <typename T, int D>
FW_DEV void store_rows16(const char* tile, char* base, long ld, int n, int wy, int wx, int H, int W, int shift, int col, int r0, int extra = 0) {
    using G = Geo<T, D>;
    const int l = lane_id();
    const int ng = G::CH + extra;
    for (int idx = l; idx < 16 * ng; idx += 64) {
        const int t = r0 + idx / ng, s = idx % ng;
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

// sum / max over the rows (j) of a [16*JT][16] score strip held as acc[jt]: registers + 2 shuffles
FW_DEV float col_reduce_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
FW_DEV float col_reduce_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// ---- B1: the disc band filter on a 64x64 tile held transposed in registers ----------------------------------
// Wave w passes / receives its strip: in[jt] = A^T[j = 16 jt ..][i = 16 w ..], out[jt] = B1(A)^T likewise.
// scrA / scrB: two workgroup-wide LDS scratch regions of >= 64*LDP and >= 2*64*LDV bytes that no wave is still reading
// on entry (the caller has passed a block barrier since their last use); on return both are free again.
template <typename T>
FW_DEV void band_filter(const f32x4 (&in)[4], f32x4 (&out)[4], char* scrA, char* scrB, const char* lfs) {
    using G = Geo<T, 56>;
    constexpr int SZ = G::SZ, LDP = G::LDP, LDV = G::LDV, JC = G::JC, VC = G::VC;
    const char* tab = lfs;
    const float* Mw = reinterpret_cast<const float*>(lfs + (size_t)OFF_END * SZ);
    const int l = lane_id(), w = wave_id();
    // Ps[i][j] <- in^T   (scrA, rows i of the own strip)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(scrA, LDP, jt * 16, w * 16, in[jt]);
    wave_fence();
    // T[i][v] = sum_j P[i][j] Fv[j][v]   (Tr with cos, Ti with -sin), rows i of the own strip; stored transposed -> Ts[v][i] (scrB)
    {
        f32x4 tr[2], ti[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { tr[n] = f32x4{0.f, 0.f, 0.f, 0.f}; ti[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < JC; ++c) {
            const uint4 a = frag_kc(scrA, LDP, w * 16, c);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint4 bc = frag_kc(tab + (size_t)OFF_C2 * SZ, 64 * SZ, n * 16, c);
                const uint4 bs = frag_kc(tab + (size_t)OFF_S2N * SZ, 64 * SZ, n * 16, c);
                mma_chunk<T>(tr[n], a, bc); mma_chunk<T>(ti[n], a, bs);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            store_acc_T<T>(scrB, LDP, w * 16, n * 16, tr[n]);                 // Tr^T [v][i]
            store_acc_T<T>(scrB + 32 * LDP, LDP, w * 16, n * 16, ti[n]);      // Ti^T [v][i]
        }
    }
    __syncthreads();                         // T complete (all strips); every wave is done reading Ps
    // X[u][v] = sum_i Fu[u][i] T[i][v]:  Xr = Cu Tr + Su Ti,  Xi = Cu Ti - Su Tr;  Y = Mw * X -> Ys[v][u] (scrA)
    // 3 row tiles of u (|fu| <= 22 -> 48 rows): waves 0..2 take one each, wave 3 clears the k padding u = 48..63 of Ys
    if (w < 3) {
        f32x4 xr[2], xi[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { xr[n] = f32x4{0.f, 0.f, 0.f, 0.f}; xi[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < JC; ++c) {
            const uint4 ac = frag_kc(tab + (size_t)OFF_CU * SZ, 64 * SZ, w * 16, c);
            const uint4 as = frag_kc(tab + (size_t)OFF_SU * SZ, 64 * SZ, w * 16, c);
            const uint4 an = frag_kc(tab + (size_t)OFF_SUN * SZ, 64 * SZ, w * 16, c);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint4 br = frag_kc(scrB, LDP, n * 16, c), bi = frag_kc(scrB + 32 * LDP, LDP, n * 16, c);
                mma_chunk<T>(xr[n], ac, br); mma_chunk<T>(xr[n], as, bi);
                mma_chunk<T>(xi[n], ac, bi); mma_chunk<T>(xi[n], an, br);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x4 wt;
#pragma unroll
            for (int r = 0; r < 4; ++r) wt[r] = Mw[(w * 16 + ((l >> 4) << 2) + r) * NV + n * 16 + (l & 15)];
            store_acc_T<T>(scrA, LDP, w * 16, n * 16, xr[n] * wt);             // Yr^T [v][u]
            store_acc_T<T>(scrA + 32 * LDP, LDP, w * 16, n * 16, xi[n] * wt);  // Yi^T [v][u]
        }
    } else {
        for (int idx = l; idx < 64 * 4; idx += 64) {
            const int row = idx >> 2, part = idx & 3;          // 64 rows (2 panels x 32), 16 pad elements in 4 parts
            char* p = scrA + row * LDP + 48 * SZ + part * 4 * SZ;
            if (SZ == 4) *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0); else *reinterpret_cast<uint2*>(p) = make_uint2(0, 0);
        }
    }
    __syncthreads();                         // Y complete; every wave is done reading T
    // Z^T[v][i] = sum_u Y^T[v][u] FuH[i][u]:  Zr = Yr C - Yi S,  Zi = Yi C + Yr S;  own columns i; stored transposed -> Zs[i][v] (scrB)
    {
        f32x4 zr[2], zi[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) { zr[m] = f32x4{0.f, 0.f, 0.f, 0.f}; zi[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < JC; ++c) {
            const uint4 bc = frag_kc(tab + (size_t)OFF_CH * SZ, 64 * SZ, w * 16, c);
            const uint4 bs = frag_kc(tab + (size_t)OFF_SH * SZ, 64 * SZ, w * 16, c);
            const uint4 bn = frag_kc(tab + (size_t)OFF_SHN * SZ, 64 * SZ, w * 16, c);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint4 ar = frag_kc(scrA, LDP, m * 16, c), ai = frag_kc(scrA + 32 * LDP, LDP, m * 16, c);
                mma_chunk<T>(zr[m], ar, bc); mma_chunk<T>(zr[m], ai, bn);
                mma_chunk<T>(zi[m], ai, bc); mma_chunk<T>(zi[m], ar, bs);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            store_acc_T<T>(scrB, LDV, m * 16, w * 16, zr[m]);                 // Zr [i][v]
            store_acc_T<T>(scrB + 64 * LDV, LDV, m * 16, w * 16, zi[m]);      // Zi [i][v]
        }
    }
    wave_fence();
    // out^T[j][i] = sum_v Gc[j][v] Zr[i][v] + Gsn[j][v] Zi[i][v]   (own columns i)
#pragma unroll
    for (int m = 0; m < 4; ++m) out[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < VC; ++c) {
        const uint4 br = frag_kc(scrB, LDV, w * 16, c), bi = frag_kc(scrB + 64 * LDV, LDV, w * 16, c);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint4 ac = frag_kc(tab + (size_t)OFF_GC * SZ, 32 * SZ, m * 16, c);
            const uint4 as = frag_kc(tab + (size_t)OFF_GSN * SZ, 32 * SZ, m * 16, c);
            mma_chunk<T>(out[m], ac, br); mma_chunk<T>(out[m], as, bi);
        }
    }
    __syncthreads();                         // both scratch regions are free again
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
__global__ __launch_bounds__(NTH, 3) void attn_fwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = Smem<T, D, NKT, LFS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* rA = smem;                        // Q -> P' (and LFS scratch A)
    char* rB = smem + S::RA;                // K tiles (kt) -> LFS scratch B -> O staging
    char* rV = rB + S::RB * NKT;            // V tiles
    const int l = lane_id(), w = wave_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    const int items = a.nwin * a.L * a.heads;
    const int i = w * 16 + (l & 15);        // the query this lane's score columns belong to
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int h = item % a.heads;
        const int lq = (item / a.heads) % a.L;
        const int win = item / (a.heads * a.L);
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        {
            TileLoad<T, D> tq, tk[NKT], tv[NKT];
            tq.issue(a.q, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int lk = a.mode == 0 ? lq : other_band(lq, kt);
                tk[kt].issue(a.k, a.ld, lk * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
                tv[kt].issue(a.v, a.ld, lk * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
            }
            tq.commit(rA);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) { tk[kt].commit(rB + kt * S::RB); tv[kt].commit(rV + kt * G::TILE_D); }
        }
        __syncthreads();
        // S^T[j][i] = sum_d K[j][d] Q[i][d], own columns i
        f32x4 p[4 * NKT];
#pragma unroll
        for (int jt = 0; jt < 4 * NKT; ++jt) p[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int c = 0; c < G::KC; ++c) {
                const uint4 bq = frag_kc(rA, G::LDR, w * 16, c);
#pragma unroll
                for (int m = 0; m < 4; ++m) mma_chunk<T>(p[4 * kt + m], frag_kc(rB + kt * S::RB, G::LDR, m * 16, c), bq);
            }
        // scale, bias, mask, softmax over j
        {
            float mx = -3.0e38f;
#pragma unroll
            for (int jt = 0; jt < 4 * NKT; ++jt) {
                const int kt = jt >> 2;
                const int lk = a.mode == 0 ? lq : other_band(lq, kt);
                const float* tab = a.bias + (size_t)(lq * a.L + lk) * 225 * a.heads;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = (jt & 3) * 16 + ((l >> 4) << 2) + r;
                    const float s = p[jt][r] * a.scale + bias_mask(tab, a.heads, h, i, j, a.shift, last_y, last_x);
                    p[jt][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
            mx = col_reduce_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4 * NKT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __expf(p[jt][r] - mx); p[jt][r] = e; sum += e; }
            sum = col_reduce_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int jt = 0; jt < 4 * NKT; ++jt) p[jt] *= inv;
            if ((l >> 4) == 0) a.lse[(size_t)item * 64 + i] = mx + __logf(sum);
        }
        __syncthreads();                    // Q / K tiles are dead from here on
        if constexpr (LFS >= 1) {
            static_assert(NKT == 1, "frequency selection acts on one 64x64 map");
            const float* cf = a.coef + ((size_t)b * a.heads + h) * 3;
            const float ca = cf[0], cb = cf[1], cc = cf[2];
            if constexpr (LFS == 2) {
                f32x4 f1[4];
                band_filter<T>(p, f1, rA, rB, a.lfs);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) p[jt] = p[jt] * ca + cb + f1[jt] * cc;
            } else {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) p[jt] = p[jt] * ca + cb;
            }
        }
        // P'[i][j] -> rA rows i of the own strip (transposed store), then O^T[d][i] = sum_j V[j][d] P'[i][j]
#pragma unroll
        for (int jt = 0; jt < 4 * NKT; ++jt) store_acc_T<T>(rA, S::LDPK, jt * 16, w * 16, p[jt]);
        wave_fence();
        f32x4 o[G::DT];
#pragma unroll
        for (int m = 0; m < G::DT; ++m) o[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int c = 0; c < G::JC; ++c) {
                const uint4 bp = frag_kc(rA, S::LDPK, w * 16, kt * G::JC + c);
#pragma unroll
                for (int m = 0; m < G::DT; ++m) mma_chunk<T>(o[m], frag_km<T>(rV + kt * G::TILE_D, G::LDR, m * 16, c), bp);
            }
#pragma unroll
        for (int m = 0; m < G::DT; ++m) store_acc_T<T>(rB, G::LDR, m * 16, w * 16, o[m]);      // O [i][d], own rows
        wave_fence();
        store_rows16<T, D>(rB, a.out, a.ldo, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
        __syncthreads();                    // before the next item's tiles overwrite what slower waves still read
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
    static constexpr int OFF_BIN = OFF_Y + SCR;                                                // float [225] (+pad)
    static constexpr int BYTES = OFF_BIN + 1024;
};

template <typename T, int D, int NKT, int LFS>
__global__ __launch_bounds__(NTH, 2) void attn_bwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = SmemB<T, D, NKT, LFS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem + S::OFF_Q; char* sDO = smem + S::OFF_DO; char* sK = smem + S::OFF_K; char* sV = smem + S::OFF_V;
    char* sX = smem + S::OFF_X; char* sY = smem + S::OFF_Y;
    const int l = lane_id(), w = wave_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    // grid: x = chunk, y = head, z = query band
    const int h = blockIdx.y, lq = blockIdx.z;
    const int dqx = (h == a.heads - 1) ? a.dq_pad * G::SZ / G::CB : 0;       // granules of zero padding behind the last head's dq
    const int i = w * 16 + (l & 15);                         // the query of this lane's score columns
    f32x4 dbacc[NKT][4];                                     // bias-gradient accumulators of (i, j = 16 jt + 4 (l>>4) + r), all windows
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dbacc[kt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A workgroup walks a CONTIGUOUS range of windows: they belong to one image (or a few), so the three coefficient-gradient
    // sums of (image, head) are carried in registers across windows and leave as ONE atomic triple per wave when the image
    // changes -- per-window atomics on the B * heads * 3 hot words serialised (same-address atomics), and every one of them
    // sat in front of the next window's tile loads in the in-order vmcnt queue.
    const int per = (a.nwin + gridDim.x - 1) / gridDim.x;
    const int win_begin = blockIdx.x * per, win_end = min(a.nwin, win_begin + per);
    float cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
    int cs_b = -1;
    auto flush_coef = [&]() {
        if constexpr (LFS >= 1) {
            if (cs_b >= 0) {
                const float t1 = wave_sum(cs1), t2 = wave_sum(cs2), t3 = wave_sum(cs3);
                if (l == 0) {
                    float* dc = a.dcoef + ((size_t)cs_b * a.heads + h) * 3;
                    atomicAdd(dc, t1); atomicAdd(dc + 1, t2);
                    if (LFS == 2) atomicAdd(dc + 2, t3);
                }
            }
            cs1 = cs2 = cs3 = 0.f;
        }
    };
    for (int win = win_begin; win < win_end; ++win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        const size_t item = ((size_t)win * a.L + lq) * a.heads + h;
        if (b != cs_b) { flush_coef(); cs_b = b; }
        TileLoad<T, D> tq, tdo;
        tq.issue(a.q, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
        tdo.issue(a.dout, a.lddo, nq, wy, wx, a.H, a.W, a.shift, h * D);
        const float lse = a.lse[item * 64 + i];
        float di = 0.f;
        if constexpr (NKT > 1) {
            // D_i = sum_d O[i][d] dO[i][d]   (rowsum(P o dP) == rowsum(O o dO) when P' = P); the 4 lanes of a column share the row
            const long row = token_row(nq, wy, wx, i, a.H, a.W, a.shift);
            const T* op = reinterpret_cast<const T*>(a.out) + row * a.ldo + h * D;
            const T* gp = reinterpret_cast<const T*>(a.dout) + row * a.lddo + h * D;
            float s = 0.f;
            for (int d = (l >> 4); d < D; d += 4) s += TT<T>::ld(op + d) * TT<T>::ld(gp + d);
            di = col_reduce_sum(s);
        }
        f32x4 dq[G::DT];
#pragma unroll
        for (int m = 0; m < G::DT; ++m) dq[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kt = 0; kt < NKT; ++kt) {
            const int lk = a.mode == 0 ? lq : other_band(lq, kt);
            const int nk = lk * a.B + b;
            const int tabid = lq * a.L + lk;
            {
                TileLoad<T, D> tk, tv;
                tk.issue(a.k, a.ld, nk, wy, wx, a.H, a.W, a.shift, h * D);
                tv.issue(a.v, a.ld, nk, wy, wx, a.H, a.W, a.shift, h * D);
                __syncthreads();                             // every wave is done with the previous K / V tiles and scratch
                if (kt == 0) { tq.commit(sQ); tdo.commit(sDO); }
                tk.commit(sK); tv.commit(sV);
            }
            __syncthreads();
            // P^T[j][i] = exp(scale * K Q^T + bias + mask - lse_i), own columns i
            f32x4 p[4], dp[4];
            const float* tab = a.bias + (size_t)tabid * 225 * a.heads;
            auto compute_p = [&]() {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) p[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < G::KC; ++c) {
                    const uint4 bq = frag_kc(sQ, G::LDR, w * 16, c);
#pragma unroll
                    for (int m = 0; m < 4; ++m) mma_chunk<T>(p[m], frag_kc(sK, G::LDR, m * 16, c), bq);
                }
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = jt * 16 + ((l >> 4) << 2) + r;
                        p[jt][r] = __expf(p[jt][r] * a.scale + bias_mask(tab, a.heads, h, i, j, a.shift, last_y, last_x) - lse);
                    }
            };
            compute_p();
            float ca = 1.f, cc = 0.f;
            // P'[i][j] = a P + b (+ c B1(P))  -> sX rows i of the own strip
            if constexpr (LFS >= 1) {
                const float* cf = a.coef + ((size_t)b * a.heads + h) * 3;
                ca = cf[0]; const float cb = cf[1]; cc = cf[2];
                if constexpr (LFS == 2) {
                    f32x4 f[4];
                    band_filter<T>(p, f, sX, sY, a.lfs);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(sX, G::LDP, jt * 16, w * 16, p[jt] * ca + cb + f[jt] * cc);
                } else {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(sX, G::LDP, jt * 16, w * 16, p[jt] * ca + cb);
                }
            } else {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(sX, G::LDP, jt * 16, w * 16, p[jt]);
            }
            __syncthreads();
            // dV^T[d][j] = sum_i dO[i][d] P'[i][j], own columns j  (both operands read along the token axis: transposing reads)
            {
                f32x4 dv[G::DT];
#pragma unroll
                for (int m = 0; m < G::DT; ++m) dv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < G::JC; ++c) {
                    const uint4 bn = frag_km<T>(sX, G::LDP, w * 16, c);
#pragma unroll
                    for (int m = 0; m < G::DT; ++m) mma_chunk<T>(dv[m], frag_km<T>(sDO, G::LDR, m * 16, c), bn);
                }
#pragma unroll
                for (int m = 0; m < G::DT; ++m) store_acc_T<T>(sY, G::LDR, m * 16, w * 16, dv[m]);   // dV [j][d], own rows j
                wave_fence();
                char* dvp = (NKT > 1 && kt_slot(lq, lk) == 1) ? a.dv2 : a.dv;
                store_rows16<T, D>(sY, dvp, a.ldd, nk, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
            }
            // dP'^T[j][i] = sum_d V[j][d] dO[i][d], own columns i
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < G::KC; ++c) {
                const uint4 bd = frag_kc(sDO, G::LDR, w * 16, c);
#pragma unroll
                for (int m = 0; m < 4; ++m) mma_chunk<T>(dp[m], frag_kc(sV, G::LDR, m * 16, c), bd);
            }
            __syncthreads();                                 // every wave is done reading P' (sX) and its dV staging rows (sY)
            if constexpr (LFS >= 1) {
                // G^T = B1(dP')^T ; d(a,b,c) = (<dP',P>, sum dP', <G,P>) ; dP = a dP' + c G
                float s1 = 0.f, s2 = 0.f, s3 = 0.f;
                if constexpr (LFS == 2) {
                    f32x4 g[4];
                    band_filter<T>(dp, g, sX, sY, a.lfs);
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s1 += dp[jt][r] * p[jt][r];
                            s2 += dp[jt][r];
                            s3 += g[jt][r] * p[jt][r];
                            dp[jt][r] = dp[jt][r] * ca + g[jt][r] * cc;
                        }
                } else {
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            s1 += dp[jt][r] * p[jt][r];
                            s2 += dp[jt][r];
                            dp[jt][r] = dp[jt][r] * ca;
                        }
                }
                cs1 += s1; cs2 += s2; cs3 += s3;
            }
            // D_i (NKT == 1: straight from registers), dS^T = P^T o (dP^T - D_i)
            if constexpr (NKT == 1) {
                float s = 0.f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += p[jt][r] * dp[jt][r];
                di = col_reduce_sum(s);
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dp[jt][r] = p[jt][r] * (dp[jt][r] - di);
                if (NKT == 1 || kt == 0) dbacc[0][jt] += dp[jt]; else dbacc[NKT - 1][jt] += dp[jt];   // static indices: registers
            }
            // dS -> sX as [i][j], rows i of the own strip; dQ reads it along j (row fragments), dK along i (transposing reads)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(sX, G::LDP, jt * 16, w * 16, dp[jt]);
            __syncthreads();
            // dQ^T[d][i] += sum_j K[j][d] dS[i][j] (own i) ;  dK^T[d][j] = sum_i Q[i][d] dS[i][j] (own j)
            f32x4 dk[G::DT];
#pragma unroll
            for (int m = 0; m < G::DT; ++m) dk[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < G::JC; ++c) {
                const uint4 bx = frag_kc(sX, G::LDP, w * 16, c), by = frag_km<T>(sX, G::LDP, w * 16, c);
#pragma unroll
                for (int m = 0; m < G::DT; ++m) {
                    mma_chunk<T>(dq[m], frag_km<T>(sK, G::LDR, m * 16, c), bx);
                    mma_chunk<T>(dk[m], frag_km<T>(sQ, G::LDR, m * 16, c), by);
                }
            }
#pragma unroll
            for (int m = 0; m < G::DT; ++m) store_acc_T<T>(sY, G::LDR, m * 16, w * 16, dk[m] * a.scale);   // dK [j][d], own rows j
            wave_fence();
            char* dkp = (NKT > 1 && kt_slot(lq, lk) == 1) ? a.dk2 : a.dk;
            store_rows16<T, D>(sY, dkp, a.ldd, nk, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
        }
        __syncthreads();                                     // sX / sY / sQ / sDO are free: all waves are past their last reads
#pragma unroll
        for (int m = 0; m < G::DT; ++m) store_acc_T<T>(sY, G::LDR, m * 16, w * 16, dq[m] * a.scale);         // dQ [i][d], own rows i
        wave_fence();
        store_rows16<T, D>(sY, a.dq, a.ldd, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16, dqx);
        // the next window's Q / dO loads touch neither sY nor anything a wave still reads
    }
    flush_coef();
    // flush the bias-gradient accumulators: fold the (i, j) pairs into the 225 relative positions on chip, then one
    // atomic per bin into the parameter layout [table][225][heads] (all workgroups of a head hit the same 225 words)
    float* bins = reinterpret_cast<float*>(smem + S::OFF_BIN);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 225; idx += NTH) bins[idx] = 0.f;
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + ((l >> 4) << 2) + r;
                atomicAdd(&bins[((i >> 3) - (j >> 3) + 7) * 15 + (i & 7) - (j & 7) + 7], dbacc[kt][jt][r]);
            }
        __syncthreads();
        const int lk = a.mode == 0 ? lq : other_band(lq, kt);
        float* dst = a.dbias + (size_t)(lq * a.L + lk) * 225 * a.heads;
        for (int idx = threadIdx.x; idx < 225; idx += NTH) atomicAdd(dst + idx * a.heads + h, bins[idx]);
    }
}


// =====================================================================================================
// v2 kernels for one key tile (decoder W-MSA + LFS, encoder intra / origin, ViT): everything a window needs beyond its own
// Q / K / V / dO tiles lives ON CHIP, and the next window's tiles are already in flight while the current one is computed.
//
// The kernels above spend their time WAITING, not computing (rocprofv3: 61 % of wave cycles in s_waitcnt; 45 us per head-window
// against 0.9 us of MFMA): per window they chain a dozen global round trips -- tile loads, the relative-position gather, eight
// rounds of DFT-panel fragments from L2, and three stores that sit in front of the next window's loads in the in-order vmcnt
// queue.  Here
//   * the DFT panels are staged once per workgroup into LDS (23 KB bf16): CH / SH / GC / GSN of the v1 table are transposes of
//     CU / SU / C2 / S2N and are read with the transposing fragment read, the negated panels are sign flips in registers; rows
//     48..63 of the u axis alias the C2 / S2N panels behind them (finite values, multiplied by the zero padding of Y);
//   * the 225 relative-position biases of the workgroup's (table, head) are staged once into LDS: a workgroup keeps one head
//     and one query band for its whole life (grid = chunks x heads x bands) and walks a contiguous range of windows;
//   * the tile loads of window t+1 are issued right after window t's tiles are committed to LDS, i.e. BEFORE window t's stores:
//     they are older in the vmcnt queue, so waiting for them never waits for a store;
//   * backward, bf16: the two applications of the band filter (on P and on dP') run as ONE pass over two tiles -- half the
//     barriers, every panel fragment used twice.
// One workgroup of 4 waves per CU (backward; LDS 111 KB) or two (forward, 77 KB): with nothing left to wait for, occupancy is not
// what hides latency any more.
template <typename T> struct FiltTab {
    static constexpr int SZ = TT<T>::SZ;
    static constexpr int LDP = 64 * SZ + 16;
    static constexpr int OFF_CU = 0, OFF_C2 = 48 * LDP, OFF_SU = 80 * LDP, OFF_S2N = 128 * LDP, OFF_MW = 160 * LDP;
    static constexpr int BYTES = OFF_MW + NU * NV * 4;      // with the mask weights on chip
    static constexpr int BYTES_NOMW = OFF_MW;                // mask weights read from global (L1-resident, 8 values per lane and window)
};
template <typename T, bool MW> FW_DEV void stage_filter_tables(char* lds, const char* lfs) {
    using F = FiltTab<T>;
    constexpr int SZ = F::SZ, GR = 64 * SZ / 16;                       // 16-byte granules per 64-element row
    for (int idx = threadIdx.x; idx < 160 * GR; idx += NTH) {
        const int r = idx / GR, g = idx % GR;
        const int src = r < 48 ? OFF_CU + r * 64 : r < 80 ? OFF_C2 + (r - 48) * 64 : r < 128 ? OFF_SU + (r - 80) * 64 : OFF_S2N + (r - 128) * 64;
        *reinterpret_cast<uint4*>(lds + r * F::LDP + g * 16) = *reinterpret_cast<const uint4*>(lfs + (size_t)src * SZ + g * 16);
    }
    if constexpr (MW) {
        const char* mw = lfs + (size_t)OFF_END * SZ;
        for (int idx = threadIdx.x; idx < NU * NV / 4; idx += NTH)
            *reinterpret_cast<uint4*>(lds + F::OFF_MW + idx * 16) = *reinterpret_cast<const uint4*>(mw + idx * 16);
    }
}
template <typename T> FW_DEV uint4 frag_neg(const uint4& v) {
    constexpr unsigned m = sizeof(T) == 2 ? 0x80008000u : 0x80000000u;
    return make_uint4(v.x ^ m, v.y ^ m, v.z ^ m, v.w ^ m);
}
// B1 on NT tiles at once.  in[t][jt] / out[t][jt] as in band_filter; scrA[t], scrB[t]: per-tile scratch pairs (>= 64*LDP and
// >= max(64*LDP, 2*64*LDV) bytes) free on entry and on return; tab: the LDS image of stage_filter_tables.
template <typename T, int NT>
FW_DEV void band_filter2(const f32x4 (&in)[NT][4], f32x4 (&out)[NT][4], char* const (&scrA)[NT], char* const (&scrB)[NT], const char* tab,
                         const float* Mw) {
    using G = Geo<T, 56>;
    using F = FiltTab<T>;
    constexpr int SZ = G::SZ, LDP = G::LDP, LDV = G::LDV, JC = G::JC, VC = G::VC;
    const int l = lane_id(), w = wave_id();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(scrA[t], LDP, jt * 16, w * 16, in[t][jt]);
    wave_fence();
    {   // T[i][v] = sum_j P[i][j] Fv[j][v]: rows i of the own strip; stored transposed -> [v][i]
        f32x4 tr[NT][2], ti[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) { tr[t][n] = f32x4{0.f, 0.f, 0.f, 0.f}; ti[t][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < JC; ++c) {
            uint4 av[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) av[t] = frag_kc(scrA[t], LDP, w * 16, c);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const uint4 bc = frag_kc(tab + F::OFF_C2, LDP, n * 16, c), bs = frag_kc(tab + F::OFF_S2N, LDP, n * 16, c);
#pragma unroll
                for (int t = 0; t < NT; ++t) { mma_chunk<T>(tr[t][n], av[t], bc); mma_chunk<T>(ti[t][n], av[t], bs); }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                store_acc_T<T>(scrB[t], LDP, w * 16, n * 16, tr[t][n]);
                store_acc_T<T>(scrB[t] + 32 * LDP, LDP, w * 16, n * 16, ti[t][n]);
            }
    }
    lds_barrier();
    if (w < 3) {   // X = Fu T, Y = Mw * X -> [v][u]; waves 0..2 own a 16-row tile of u each, wave 3 clears the k padding u = 48..63
        f32x4 xr[NT][2], xi[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) { xr[t][n] = f32x4{0.f, 0.f, 0.f, 0.f}; xi[t][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < JC; ++c) {
            const uint4 ac = frag_kc(tab + F::OFF_CU, LDP, w * 16, c), as = frag_kc(tab + F::OFF_SU, LDP, w * 16, c);
            const uint4 an = frag_neg<T>(as);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const uint4 br = frag_kc(scrB[t], LDP, n * 16, c), bi = frag_kc(scrB[t] + 32 * LDP, LDP, n * 16, c);
                    mma_chunk<T>(xr[t][n], ac, br); mma_chunk<T>(xr[t][n], as, bi);
                    mma_chunk<T>(xi[t][n], ac, bi); mma_chunk<T>(xi[t][n], an, br);
                }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x4 wt;
#pragma unroll
            for (int r = 0; r < 4; ++r) wt[r] = Mw[(w * 16 + ((l >> 4) << 2) + r) * NV + n * 16 + (l & 15)];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                store_acc_T<T>(scrA[t], LDP, w * 16, n * 16, xr[t][n] * wt);
                store_acc_T<T>(scrA[t] + 32 * LDP, LDP, w * 16, n * 16, xi[t][n] * wt);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            for (int idx = l; idx < 64 * 4; idx += 64) {
                const int row = idx >> 2, part = idx & 3;
                char* p = scrA[t] + row * LDP + 48 * SZ + part * 4 * SZ;
                if (SZ == 4) *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0); else *reinterpret_cast<uint2*>(p) = make_uint2(0, 0);
            }
    }
    lds_barrier();
    {   // Z^T[v][i] = sum_u Y^T[v][u] FuH[i][u] (own columns i); FuH = transposed CU / SU panels; stored transposed -> [i][v]
        f32x4 zr[NT][2], zi[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m) { zr[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}; zi[t][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < JC; ++c) {
            const uint4 bc = frag_km<T>(tab + F::OFF_CU, LDP, w * 16, c), bs = frag_km<T>(tab + F::OFF_SU, LDP, w * 16, c);
            const uint4 bn = frag_neg<T>(bs);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const uint4 ar = frag_kc(scrA[t], LDP, m * 16, c), ai = frag_kc(scrA[t] + 32 * LDP, LDP, m * 16, c);
                    mma_chunk<T>(zr[t][m], ar, bc); mma_chunk<T>(zr[t][m], ai, bn);
                    mma_chunk<T>(zi[t][m], ai, bc); mma_chunk<T>(zi[t][m], ar, bs);
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                store_acc_T<T>(scrB[t], LDV, m * 16, w * 16, zr[t][m]);
                store_acc_T<T>(scrB[t] + 64 * LDV, LDV, m * 16, w * 16, zi[t][m]);
            }
    }
    wave_fence();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) out[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < VC; ++c) {   // out^T[j][i] = sum_v G[j][v] Z[i][v]; G = transposed C2 / S2N panels
        uint4 ac[4], as[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { ac[m] = frag_km<T>(tab + F::OFF_C2, LDP, m * 16, c); as[m] = frag_km<T>(tab + F::OFF_S2N, LDP, m * 16, c); }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint4 br = frag_kc(scrB[t], LDV, w * 16, c), bi = frag_kc(scrB[t] + 64 * LDV, LDV, w * 16, c);
#pragma unroll
            for (int m = 0; m < 4; ++m) { mma_chunk<T>(out[t][m], ac[m], br); mma_chunk<T>(out[t][m], as[m], bi); }
        }
    }
    lds_barrier();
}

// bias + shift mask with the 225 biases of (table, head) in LDS
FW_DEV float bias_mask_lds(const float* bins, int i, int j, int shift, bool last_y, bool last_x) {
    float b = bins[((i >> 3) - (j >> 3) + 7) * 15 + (i & 7) - (j & 7) + 7];
    if (shift > 0) {
        const int s = 8 - shift;
        const int ri = (last_y ? ((i >> 3) < s ? 1 : 2) : 0) * 3 + (last_x ? ((i & 7) < s ? 1 : 2) : 0);
        const int rj = (last_y ? ((j >> 3) < s ? 1 : 2) : 0) * 3 + (last_x ? ((j & 7) < s ? 1 : 2) : 0);
        if (ri != rj) b += -100.0f;
    }
    return b;
}

// DUAL = true : both backward filters in one pass, mask weights on chip, one workgroup per CU (bf16: 111 KB of LDS)
// DUAL = false: one filter at a time, mask weights from global memory: 79.5 KB -> two workgroups per CU (bf16)
// The 16 score elements of a lane are the same (query i, key j) pairs in every window: their relative-position biases and
// whether the pair straddles the cyclic shift's seam vertically / horizontally are computed ONCE per workgroup; per window the
// -100 mask is one select per element (it applies in the last window row / column only, decoder_Uformer.py:634-651).
struct BiasRegs {
    f32x4 b[4];
    unsigned dy, dx;                                    // bit (jt * 4 + r): i and j lie on different sides of the seam
    FW_MEM void init(const float* bins, int i, int shift) {
        const int l = lane_id();
        dy = dx = 0;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + ((l >> 4) << 2) + r;
                b[jt][r] = bins[((i >> 3) - (j >> 3) + 7) * 15 + (i & 7) - (j & 7) + 7];
                if (shift > 0) {
                    const int s = 8 - shift;
                    if (((i >> 3) < s) != ((j >> 3) < s)) dy |= 1u << (jt * 4 + r);
                    if (((i & 7) < s) != ((j & 7) < s)) dx |= 1u << (jt * 4 + r);
                }
            }
    }
    FW_MEM float get(int jt, int r, unsigned m) const { return b[jt][r] + (((m >> (jt * 4 + r)) & 1) ? -100.0f : 0.0f); }
    FW_MEM unsigned mask(bool last_y, bool last_x) const { return (last_y ? dy : 0u) | (last_x ? dx : 0u); }
};

template <typename T, int D, int LFS, bool DUAL = true> struct Smem2 {
    using G = Geo<T, D>;
    static constexpr bool MW = DUAL || sizeof(T) == 4;
    static constexpr int TAB = LFS == 2 ? (MW ? FiltTab<T>::BYTES : FiltTab<T>::BYTES_NOMW) : 0;
    static constexpr int SCR = 2 * 64 * G::LDV > 64 * G::LDP ? 2 * 64 * G::LDV : 64 * G::LDP;
    static constexpr int NTF = (LFS == 2 && sizeof(T) == 2 && DUAL) ? 2 : 1;         // tiles per backward filter pass
    static constexpr int OFF_BIAS = TAB;
    static constexpr int OFF_T = TAB + 1024;                                          // tiles start here
    static constexpr int RA = SCR > G::TILE_D ? SCR : G::TILE_D;                      // fwd: Q -> P' / scratch A
    static constexpr int FWD_BYTES = OFF_T + RA + RA + G::TILE_D;                     // | A | B (K, scratch B, O staging) | V
    static constexpr int BWD_BYTES = OFF_T + 4 * G::TILE_D + 2 * NTF * SCR;           // Q dO K V | (X, Y) per filter tile
};

template <typename T, int D, int LFS>
__global__ __launch_bounds__(NTH, 1) void attn2_fwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = Smem2<T, D, LFS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* tab = smem;
    float* bins = reinterpret_cast<float*>(smem + S::OFF_BIAS);
    char* rA = smem + S::OFF_T; char* rB = rA + S::RA; char* rV = rB + S::RA;
    const int l = lane_id(), w = wave_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    const int h = blockIdx.y, lq = blockIdx.z;
    const int i = w * 16 + (l & 15);
    const int per = (a.nwin + gridDim.x - 1) / gridDim.x;
    const int win_begin = blockIdx.x * per, win_end = min(a.nwin, win_begin + per);
    if (win_begin >= win_end) return;
    if constexpr (LFS == 2) stage_filter_tables<T, true>(smem, a.lfs);
    const float* Mw = reinterpret_cast<const float*>(smem + FiltTab<T>::OFF_MW);
    {
        const float* tb = a.bias + (size_t)(lq * a.L + lq) * 225 * a.heads + h;
        for (int idx = threadIdx.x; idx < 225; idx += NTH) bins[idx] = tb[(size_t)idx * a.heads];
    }
    TileLoad<T, D> tq, tk, tv;
    auto issue = [&](int win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const int nq = lq * a.B + b;
        tq.issue(a.q, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
        tk.issue(a.k, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
        tv.issue(a.v, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
    };
    issue(win_begin);
    lds_barrier();                          // biases staged
    BiasRegs br;
    br.init(bins, i, a.shift);
    float ca = 1.f, cb = 0.f, cc = 0.f;     // the (image, head)'s LFS coefficients: re-read only when the image changes
    int cf_b = -1;
    for (int win = win_begin; win < win_end; ++win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        const size_t item = ((size_t)win * a.L + lq) * a.heads + h;
        lds_barrier();                    // the previous window's readers are done (also: tables / biases are staged)
        tq.commit(rA); tk.commit(rB); tv.commit(rV);
        lds_barrier();
        if (win + 1 < win_end) issue(win + 1);      // in flight during this window's compute, older than its stores
        f32x4 p[1][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) p[0][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < G::KC; ++c) {
            const uint4 bq = frag_kc(rA, G::LDR, w * 16, c);
#pragma unroll
            for (int m = 0; m < 4; ++m) mma_chunk<T>(p[0][m], frag_kc(rB, G::LDR, m * 16, c), bq);
        }
        {
            float mx = -3.0e38f;
            const unsigned msk = br.mask(last_y, last_x);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = p[0][jt][r] * a.scale + br.get(jt, r, msk);
                    p[0][jt][r] = sc;
                    mx = fmaxf(mx, sc);
                }
            mx = col_reduce_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __expf(p[0][jt][r] - mx); p[0][jt][r] = e; sum += e; }
            sum = col_reduce_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) p[0][jt] *= inv;
            if ((l >> 4) == 0) a.lse[item * 64 + i] = mx + __logf(sum);
        }
        lds_barrier();                    // Q / K tiles are dead from here on
        if constexpr (LFS >= 1) {
            if (b != cf_b) { const float* cf = a.coef + ((size_t)b * a.heads + h) * 3; ca = cf[0]; cb = cf[1]; cc = cf[2]; cf_b = b; }
            if constexpr (LFS == 2) {
                f32x4 f1[1][4];
                char* const sa[1] = {rA}; char* const sb[1] = {rB};
                band_filter2<T, 1>(p, f1, sa, sb, tab, Mw);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) p[0][jt] = p[0][jt] * ca + cb + f1[0][jt] * cc;
            } else {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) p[0][jt] = p[0][jt] * ca + cb;
            }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(rA, G::LDP, jt * 16, w * 16, p[0][jt]);
        wave_fence();
        f32x4 o[G::DT];
#pragma unroll
        for (int m = 0; m < G::DT; ++m) o[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < G::JC; ++c) {
            const uint4 bp = frag_kc(rA, G::LDP, w * 16, c);
#pragma unroll
            for (int m = 0; m < G::DT; ++m) mma_chunk<T>(o[m], frag_km<T>(rV, G::LDR, m * 16, c), bp);
        }
#pragma unroll
        for (int m = 0; m < G::DT; ++m) store_acc_T<T>(rB, G::LDR, m * 16, w * 16, o[m]);
        wave_fence();
        store_rows16<T, D>(rB, a.out, a.ldo, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
    }
}

template <typename T, int D, int LFS, bool DUAL>
__global__ __launch_bounds__(NTH, 1) void attn2_bwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = Smem2<T, D, LFS, DUAL>;
    constexpr int NTF = S::NTF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* tab = smem;
    float* bins = reinterpret_cast<float*>(smem + S::OFF_BIAS);
    char* sQ = smem + S::OFF_T; char* sDO = sQ + G::TILE_D; char* sK = sDO + G::TILE_D; char* sV = sK + G::TILE_D;
    char* sX = sV + G::TILE_D; char* sY = sX + S::SCR;
    char* sX2 = NTF == 2 ? sY + S::SCR : sX; char* sY2 = NTF == 2 ? sX2 + S::SCR : sY;
    const int l = lane_id(), w = wave_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    const int h = blockIdx.y, lq = blockIdx.z;
    const int dqx = (h == a.heads - 1) ? a.dq_pad * G::SZ / G::CB : 0;       // granules of zero padding behind the last head's dq
    const int i = w * 16 + (l & 15);
    const int per = (a.nwin + gridDim.x - 1) / gridDim.x;
    const int win_begin = blockIdx.x * per, win_end = min(a.nwin, win_begin + per);
    if (win_begin >= win_end) return;
    if constexpr (LFS == 2) stage_filter_tables<T, S::MW>(smem, a.lfs);
    const float* Mw = S::MW ? reinterpret_cast<const float*>(smem + FiltTab<T>::OFF_MW)
                            : reinterpret_cast<const float*>(a.lfs + (size_t)OFF_END * TT<T>::SZ);
    {
        const float* tb = a.bias + (size_t)(lq * a.L + lq) * 225 * a.heads + h;
        for (int idx = threadIdx.x; idx < 225; idx += NTH) bins[idx] = tb[(size_t)idx * a.heads];
    }
    f32x4 dbacc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) dbacc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs1 = 0.f, cs2 = 0.f, cs3 = 0.f, ca = 1.f, cb = 0.f, cc = 0.f;
    int cs_b = -1;
    auto flush_coef = [&]() {
        if constexpr (LFS >= 1) {
            if (cs_b >= 0) {
                const float t1 = wave_sum(cs1), t2 = wave_sum(cs2), t3 = wave_sum(cs3);
                if (l == 0) {
                    float* dc = a.dcoef + ((size_t)cs_b * a.heads + h) * 3;
                    atomicAdd(dc, t1); atomicAdd(dc + 1, t2);
                    if (LFS == 2) atomicAdd(dc + 2, t3);
                }
            }
            cs1 = cs2 = cs3 = 0.f;
        }
    };
    TileLoad<T, D> tq, tdo, tk, tv;
    auto issue = [&](int win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const int nq = lq * a.B + b;
        tq.issue(a.q, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
        tdo.issue(a.dout, a.lddo, nq, wy, wx, a.H, a.W, a.shift, h * D);
        tk.issue(a.k, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
        tv.issue(a.v, a.ld, nq, wy, wx, a.H, a.W, a.shift, h * D);
    };
    issue(win_begin);
    lds_barrier();                          // biases staged
    BiasRegs br;
    br.init(bins, i, a.shift);
    float lse_next = a.lse[(((size_t)win_begin * a.L + lq) * a.heads + h) * 64 + i];
    for (int win = win_begin; win < win_end; ++win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        if (b != cs_b) {
            flush_coef(); cs_b = b;
            if constexpr (LFS >= 1) { const float* cf = a.coef + ((size_t)b * a.heads + h) * 3; ca = cf[0]; cb = cf[1]; cc = cf[2]; }
        }
        const float lse = lse_next;
        lds_barrier();                    // every wave is past its last read of the previous window's tiles / scratch
        tq.commit(sQ); tdo.commit(sDO); tk.commit(sK); tv.commit(sV);
        lds_barrier();
        if (win + 1 < win_end) {
            issue(win + 1);
            lse_next = a.lse[(((size_t)(win + 1) * a.L + lq) * a.heads + h) * 64 + i];
        }
        // P^T[j][i] = exp(scale K Q^T + bias + mask - lse_i), dP'^T[j][i] = V dO^T  -- own columns i; both need only the tiles
        f32x4 pd[2][4];                      // [0] = P, [1] = dP'
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) { pd[0][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; pd[1][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < G::KC; ++c) {
            const uint4 bq = frag_kc(sQ, G::LDR, w * 16, c), bd = frag_kc(sDO, G::LDR, w * 16, c);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                mma_chunk<T>(pd[0][m], frag_kc(sK, G::LDR, m * 16, c), bq);
                mma_chunk<T>(pd[1][m], frag_kc(sV, G::LDR, m * 16, c), bd);
            }
        }
        const unsigned msk = br.mask(last_y, last_x);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pd[0][jt][r] = __expf(pd[0][jt][r] * a.scale + br.get(jt, r, msk) - lse);
        f32x4 fg[2][4];                      // [0] = B1(P), [1] = B1(dP')
        if constexpr (LFS == 2) {
            if constexpr (NTF == 2) {
                char* const sa[2] = {sX, sX2}; char* const sb[2] = {sY, sY2};
                band_filter2<T, 2>(pd, fg, sa, sb, tab, Mw);
            } else {
                char* const sa[1] = {sX}; char* const sb[1] = {sY};
                f32x4 t0[1][4], t1[1][4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) t0[0][jt] = pd[0][jt];
                band_filter2<T, 1>(t0, t1, sa, sb, tab, Mw);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) { fg[0][jt] = t1[0][jt]; t0[0][jt] = pd[1][jt]; }
                band_filter2<T, 1>(t0, t1, sa, sb, tab, Mw);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) fg[1][jt] = t1[0][jt];
            }
        }
        // P' = a P + b (+ c B1(P)) -> sX rows i of the own strip
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 pp = pd[0][jt] * ca + cb;
            if constexpr (LFS == 2) pp += fg[0][jt] * cc;
            store_acc_T<T>(sX, G::LDP, jt * 16, w * 16, LFS >= 1 ? pp : pd[0][jt]);
        }
        lds_barrier();
        {   // dV^T[d][j] = sum_i dO[i][d] P'[i][j], own columns j
            f32x4 dv[G::DT];
#pragma unroll
            for (int m = 0; m < G::DT; ++m) dv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < G::JC; ++c) {
                const uint4 bn = frag_km<T>(sX, G::LDP, w * 16, c);
#pragma unroll
                for (int m = 0; m < G::DT; ++m) mma_chunk<T>(dv[m], frag_km<T>(sDO, G::LDR, m * 16, c), bn);
            }
#pragma unroll
            for (int m = 0; m < G::DT; ++m) store_acc_T<T>(sY, G::LDR, m * 16, w * 16, dv[m]);
            wave_fence();
            store_rows16<T, D>(sY, a.dv, a.ldd, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
        }
        // d(a, b, c) = (<dP', P>, sum dP', <B1(dP'), P>);  dP = a dP' + c B1(dP');  D_i;  dS = P o (dP - D_i)
        {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f, sd = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float dpv = pd[1][jt][r];
                    if constexpr (LFS >= 1) {
                        s1 += dpv * pd[0][jt][r]; s2 += dpv;
                        if constexpr (LFS == 2) { s3 += fg[1][jt][r] * pd[0][jt][r]; dpv = dpv * ca + fg[1][jt][r] * cc; }
                        else dpv *= ca;
                    }
                    pd[1][jt][r] = dpv;
                    sd += pd[0][jt][r] * dpv;
                }
            cs1 += s1; cs2 += s2; cs3 += s3;
            const float di = col_reduce_sum(sd);
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pd[1][jt][r] = pd[0][jt][r] * (pd[1][jt][r] - di);
                dbacc[jt] += pd[1][jt];
            }
        }
        lds_barrier();                    // every wave is done reading P' (sX) and its dV staging rows (sY)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(sX, G::LDP, jt * 16, w * 16, pd[1][jt]);
        lds_barrier();
        // dQ^T[d][i] = sum_j K[j][d] dS[i][j] (own i);  dK^T[d][j] = sum_i Q[i][d] dS[i][j] (own j)
        f32x4 dq[G::DT], dk[G::DT];
#pragma unroll
        for (int m = 0; m < G::DT; ++m) { dq[m] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < G::JC; ++c) {
            const uint4 bx = frag_kc(sX, G::LDP, w * 16, c), by = frag_km<T>(sX, G::LDP, w * 16, c);
#pragma unroll
            for (int m = 0; m < G::DT; ++m) {
                mma_chunk<T>(dq[m], frag_km<T>(sK, G::LDR, m * 16, c), bx);
                mma_chunk<T>(dk[m], frag_km<T>(sQ, G::LDR, m * 16, c), by);
            }
        }
        // own rows only: dK -> sY (rows j), dQ -> sV (rows i; V is dead since dP' was formed) -- no barrier between the two stores
#pragma unroll
        for (int m = 0; m < G::DT; ++m) {
            store_acc_T<T>(sY, G::LDR, m * 16, w * 16, dk[m] * a.scale);
            store_acc_T<T>(sV, G::LDR, m * 16, w * 16, dq[m] * a.scale);
        }
        wave_fence();
        store_rows16<T, D>(sY, a.dk, a.ldd, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
        store_rows16<T, D>(sV, a.dq, a.ldd, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16, dqx);
    }
    flush_coef();
    lds_barrier();
    float* fold = reinterpret_cast<float*>(smem + S::OFF_T);          // tiles are dead: fold the (i, j) pairs into the 225 relative positions
    for (int idx = threadIdx.x; idx < 225; idx += NTH) fold[idx] = 0.f;
    lds_barrier();
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jt * 16 + ((l >> 4) << 2) + r;
            atomicAdd(&fold[((i >> 3) - (j >> 3) + 7) * 15 + (i & 7) - (j & 7) + 7], dbacc[jt][r]);
        }
    lds_barrier();
    float* dst = a.dbias + (size_t)(lq * a.L + lq) * 225 * a.heads;
    for (int idx = threadIdx.x; idx < 225; idx += NTH) atomicAdd(dst + idx * a.heads + h, fold[idx]);
}

// =====================================================================================================
// v2 scheme for INTER-band attention (mode 1, L = 3: the queries of band lq over the 2 x 64 keys of the other two bands,
// encoder_Uformer.py inter-frequency blocks; no frequency selection there).  Same structure as attn2_*: a workgroup walks a
// contiguous range of windows of one (head, query band); the 225 biases of BOTH (lq, lk) tables sit in LDS and each lane's 2 x 16
// of them in registers; the five (fwd) / six (bwd) tiles of window w + 1 are requested before window w is computed and stored;
// barriers are LDS-only.  The v1 kernel walked the two key bands one after the other through ONE K / V buffer -- per window two
// exposed load round trips, plus a global read of O and dO for D_i that the joint form gets from P and dP in registers.
// =====================================================================================================
template <typename T, int D> struct Smem2X {
    using G = Geo<T, D>;
    static constexpr int LDPK = 128 * G::SZ + 16;                           // row stride of a [64 queries][2 x 64 keys] tile
    static constexpr int PB = 64 * LDPK;
    static constexpr int OFF_T = 2048;                                      // float bins[2][256] in front
    static constexpr int RQ = PB > G::TILE_D ? PB : G::TILE_D;              // fwd: Q -> P
    static constexpr int FWD_BYTES = OFF_T + RQ + 4 * G::TILE_D;            // | Q / P | K0 K1 (O staging) | V0 V1 |
    static constexpr int BWD_BYTES = OFF_T + 8 * G::TILE_D + PB;            // | Q dO K0 K1 V0 V1 | X = P / dS | Y0 Y1 staging |
};

template <typename T, int D>
__global__ __launch_bounds__(NTH, 1) void attn2x_fwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = Smem2X<T, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bins = reinterpret_cast<float*>(smem);
    char* rA = smem + S::OFF_T; char* rK = rA + S::RQ; char* rV = rK + 2 * G::TILE_D;
    const int l = lane_id(), w = wave_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    const int h = blockIdx.y, lq = blockIdx.z;
    const int i = w * 16 + (l & 15);
    const int per = (a.nwin + gridDim.x - 1) / gridDim.x;
    const int win_begin = blockIdx.x * per, win_end = min(a.nwin, win_begin + per);
    if (win_begin >= win_end) return;
    const int lk[2] = {other_band(lq, 0), other_band(lq, 1)};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const float* tb = a.bias + (size_t)(lq * a.L + lk[kt]) * 225 * a.heads + h;
        for (int idx = threadIdx.x; idx < 225; idx += NTH) bins[kt * 256 + idx] = tb[(size_t)idx * a.heads];
    }
    TileLoad<T, D> tq, tk[2], tv[2];
    auto issue = [&](int win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        tq.issue(a.q, a.ld, lq * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            tk[kt].issue(a.k, a.ld, lk[kt] * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
            tv[kt].issue(a.v, a.ld, lk[kt] * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
        }
    };
    issue(win_begin);
    lds_barrier();                          // biases staged
    BiasRegs br[2];
    br[0].init(bins, i, a.shift);
    br[1].init(bins + 256, i, a.shift);
    for (int win = win_begin; win < win_end; ++win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        const size_t item = ((size_t)win * a.L + lq) * a.heads + h;
        lds_barrier();                    // the previous window's readers are done
        tq.commit(rA);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { tk[kt].commit(rK + kt * G::TILE_D); tv[kt].commit(rV + kt * G::TILE_D); }
        lds_barrier();
        if (win + 1 < win_end) issue(win + 1);
        f32x4 p[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) p[kt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < G::KC; ++c) {
            const uint4 bq = frag_kc(rA, G::LDR, w * 16, c);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int m = 0; m < 4; ++m) mma_chunk<T>(p[kt][m], frag_kc(rK + kt * G::TILE_D, G::LDR, m * 16, c), bq);
        }
        {
            float mx = -3.0e38f;
            const unsigned msk = br[0].mask(last_y, last_x);          // the seam flags depend on (i, j) only: the same for both key bands
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = p[kt][jt][r] * a.scale + br[kt].get(jt, r, msk);
                        p[kt][jt][r] = sc;
                        mx = fmaxf(mx, sc);
                    }
            mx = col_reduce_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float e = __expf(p[kt][jt][r] - mx); p[kt][jt][r] = e; sum += e; }
            sum = col_reduce_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) p[kt][jt] *= inv;
            if ((l >> 4) == 0) a.lse[item * 64 + i] = mx + __logf(sum);
        }
        lds_barrier();                    // Q / K tiles are dead from here on
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(rA, S::LDPK, kt * 64 + jt * 16, w * 16, p[kt][jt]);
        wave_fence();
        f32x4 o[G::DT];
#pragma unroll
        for (int m = 0; m < G::DT; ++m) o[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int c = 0; c < G::JC; ++c) {
                const uint4 bp = frag_kc(rA, S::LDPK, w * 16, kt * G::JC + c);
#pragma unroll
                for (int m = 0; m < G::DT; ++m) mma_chunk<T>(o[m], frag_km<T>(rV + kt * G::TILE_D, G::LDR, m * 16, c), bp);
            }
#pragma unroll
        for (int m = 0; m < G::DT; ++m) store_acc_T<T>(rK, G::LDR, m * 16, w * 16, o[m]);
        wave_fence();
        store_rows16<T, D>(rK, a.out, a.ldo, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
    }
}

template <typename T, int D>
__global__ __launch_bounds__(NTH, 1) void attn2x_bwd_kernel(AttnArgs a) {
    using G = Geo<T, D>;
    using S = Smem2X<T, D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bins = reinterpret_cast<float*>(smem);
    char* sQ = smem + S::OFF_T; char* sDO = sQ + G::TILE_D; char* sK = sDO + G::TILE_D; char* sV = sK + 2 * G::TILE_D;
    char* sX = sV + 2 * G::TILE_D; char* sY = sX + S::PB;
    const int l = lane_id(), w = wave_id();
    const int nWx = a.W / 8, nWy = a.H / 8, nW = nWx * nWy;
    const int h = blockIdx.y, lq = blockIdx.z;
    const int dqx = (h == a.heads - 1) ? a.dq_pad * G::SZ / G::CB : 0;       // granules of zero padding behind the last head's dq
    const int i = w * 16 + (l & 15);
    const int per = (a.nwin + gridDim.x - 1) / gridDim.x;
    const int win_begin = blockIdx.x * per, win_end = min(a.nwin, win_begin + per);
    if (win_begin >= win_end) return;
    const int lk[2] = {other_band(lq, 0), other_band(lq, 1)};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const float* tb = a.bias + (size_t)(lq * a.L + lk[kt]) * 225 * a.heads + h;
        for (int idx = threadIdx.x; idx < 225; idx += NTH) bins[kt * 256 + idx] = tb[(size_t)idx * a.heads];
    }
    f32x4 dbacc[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dbacc[kt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    TileLoad<T, D> tq, tdo, tk[2], tv[2];
    auto issue = [&](int win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        tq.issue(a.q, a.ld, lq * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
        tdo.issue(a.dout, a.lddo, lq * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            tk[kt].issue(a.k, a.ld, lk[kt] * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
            tv[kt].issue(a.v, a.ld, lk[kt] * a.B + b, wy, wx, a.H, a.W, a.shift, h * D);
        }
    };
    issue(win_begin);
    lds_barrier();                          // biases staged
    BiasRegs br[2];
    br[0].init(bins, i, a.shift);
    br[1].init(bins + 256, i, a.shift);
    float lse_next = a.lse[(((size_t)win_begin * a.L + lq) * a.heads + h) * 64 + i];
    for (int win = win_begin; win < win_end; ++win) {
        const int b = win / nW, wi = win % nW, wy = wi / nWx, wx = wi % nWx;
        const bool last_y = wy == nWy - 1, last_x = wx == nWx - 1;
        const int nq = lq * a.B + b;
        const float lse = lse_next;
        lds_barrier();                    // every wave is past its last read of the previous window's tiles / staging
        tq.commit(sQ); tdo.commit(sDO);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { tk[kt].commit(sK + kt * G::TILE_D); tv[kt].commit(sV + kt * G::TILE_D); }
        lds_barrier();
        if (win + 1 < win_end) {
            issue(win + 1);
            lse_next = a.lse[(((size_t)(win + 1) * a.L + lq) * a.heads + h) * 64 + i];
        }
        // P^T[j][i] = exp(scale K Q^T + bias + mask - lse_i), dP^T[j][i] = V dO^T -- own columns i, both key bands
        f32x4 P[2][4], dP[2][4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) { P[kt][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dP[kt][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int c = 0; c < G::KC; ++c) {
            const uint4 bq = frag_kc(sQ, G::LDR, w * 16, c), bd = frag_kc(sDO, G::LDR, w * 16, c);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    mma_chunk<T>(P[kt][m], frag_kc(sK + kt * G::TILE_D, G::LDR, m * 16, c), bq);
                    mma_chunk<T>(dP[kt][m], frag_kc(sV + kt * G::TILE_D, G::LDR, m * 16, c), bd);
                }
        }
        const unsigned msk = br[0].mask(last_y, last_x);
        float sd = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __expf(P[kt][jt][r] * a.scale + br[kt].get(jt, r, msk) - lse);
                    P[kt][jt][r] = pv;
                    sd += pv * dP[kt][jt][r];
                }
                store_acc_T<T>(sX, S::LDPK, kt * 64 + jt * 16, w * 16, P[kt][jt]);      // P [i][j], own rows i
            }
        lds_barrier();
        {   // dV_kt^T[d][j] = sum_i dO[i][d] P[i][j], own columns j of each key band
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4 dv[G::DT];
#pragma unroll
                for (int m = 0; m < G::DT; ++m) dv[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < G::JC; ++c) {
                    const uint4 bn = frag_km<T>(sX, S::LDPK, kt * 64 + w * 16, c);
#pragma unroll
                    for (int m = 0; m < G::DT; ++m) mma_chunk<T>(dv[m], frag_km<T>(sDO, G::LDR, m * 16, c), bn);
                }
#pragma unroll
                for (int m = 0; m < G::DT; ++m) store_acc_T<T>(sY + kt * G::TILE_D, G::LDR, m * 16, w * 16, dv[m]);
            }
            wave_fence();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                char* dvp = kt_slot(lq, lk[kt]) == 1 ? a.dv2 : a.dv;
                store_rows16<T, D>(sY + kt * G::TILE_D, dvp, a.ldd, lk[kt] * a.B + b, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
            }
        }
        // D_i = sum_j P dP over both bands;  dS = P o (dP - D_i)
        {
            const float di = col_reduce_sum(sd);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dP[kt][jt][r] = P[kt][jt][r] * (dP[kt][jt][r] - di);
                    dbacc[kt][jt] += dP[kt][jt];
                }
        }
        lds_barrier();                    // every wave is done reading P (sX) and its dV staging rows (sY)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) store_acc_T<T>(sX, S::LDPK, kt * 64 + jt * 16, w * 16, dP[kt][jt]);
        lds_barrier();
        // dQ^T[d][i] = sum_kt sum_j K_kt[j][d] dS[i][j] (own i);  dK_kt^T[d][j] = sum_i Q[i][d] dS[i][j] (own j)
        f32x4 dq[G::DT], dk[2][G::DT];
#pragma unroll
        for (int m = 0; m < G::DT; ++m) { dq[m] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[0][m] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[1][m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int c = 0; c < G::JC; ++c) {
                const uint4 bx = frag_kc(sX, S::LDPK, w * 16, kt * G::JC + c), by = frag_km<T>(sX, S::LDPK, kt * 64 + w * 16, c);
#pragma unroll
                for (int m = 0; m < G::DT; ++m) {
                    mma_chunk<T>(dq[m], frag_km<T>(sK + kt * G::TILE_D, G::LDR, m * 16, c), bx);
                    mma_chunk<T>(dk[kt][m], frag_km<T>(sQ, G::LDR, m * 16, c), by);
                }
            }
        // own rows only: dK_kt -> sY tiles (rows j), dQ -> sV tile 0 (rows i; V is dead since dP was formed)
#pragma unroll
        for (int m = 0; m < G::DT; ++m) {
            store_acc_T<T>(sY, G::LDR, m * 16, w * 16, dk[0][m] * a.scale);
            store_acc_T<T>(sY + G::TILE_D, G::LDR, m * 16, w * 16, dk[1][m] * a.scale);
            store_acc_T<T>(sV, G::LDR, m * 16, w * 16, dq[m] * a.scale);
        }
        wave_fence();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            char* dkp = kt_slot(lq, lk[kt]) == 1 ? a.dk2 : a.dk;
            store_rows16<T, D>(sY + kt * G::TILE_D, dkp, a.ldd, lk[kt] * a.B + b, wy, wx, a.H, a.W, a.shift, h * D, w * 16);
        }
        store_rows16<T, D>(sV, a.dq, a.ldd, nq, wy, wx, a.H, a.W, a.shift, h * D, w * 16, dqx);
    }
    float* fold = reinterpret_cast<float*>(smem + S::OFF_T);          // tiles are dead: fold the (i, j) pairs into the 225 relative positions
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        lds_barrier();
        for (int idx = threadIdx.x; idx < 225; idx += NTH) fold[idx] = 0.f;
        lds_barrier();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt * 16 + ((l >> 4) << 2) + r;
                atomicAdd(&fold[((i >> 3) - (j >> 3) + 7) * 15 + (i & 7) - (j & 7) + 7], dbacc[kt][jt][r]);
            }
        lds_barrier();
        float* dst = a.dbias + (size_t)(lq * a.L + lk[kt]) * 225 * a.heads;
        for (int idx = threadIdx.x; idx < 225; idx += NTH) atomicAdd(dst + idx * a.heads + h, fold[idx]);
    }
}

template <typename T, int D>
int x2_launch(bool bwd, const AttnArgs& a, hipStream_t st) {
    using S = Smem2X<T, D>;
    const int bytes = bwd ? S::BWD_BYTES : S::FWD_BYTES;
    if (bwd) { FW_SET_LDS_ONCE((attn2x_bwd_kernel<T, D>), S::BWD_BYTES); } else { FW_SET_LDS_ONCE((attn2x_fwd_kernel<T, D>), S::FWD_BYTES); }
    const int per_cu = 2 * bytes <= 160 * 1024 ? 2 : 1;
    int chunks = (256 * per_cu) / (a.heads * a.L);
    if (chunks < 1) chunks = 1;
    if (chunks > a.nwin) chunks = a.nwin;
    if (bwd) hipLaunchKernelGGL((attn2x_bwd_kernel<T, D>), dim3(chunks, a.heads, a.L), dim3(NTH), bytes, st, a);
    else hipLaunchKernelGGL((attn2x_fwd_kernel<T, D>), dim3(chunks, a.heads, a.L), dim3(NTH), bytes, st, a);
    FW_LAUNCH_RET();
}

template <typename T, int D, int LFS>
int fwd2_launch(const AttnArgs& a, hipStream_t st) {
    using S = Smem2<T, D, LFS>;
    FW_SET_LDS_ONCE((attn2_fwd_kernel<T, D, LFS>), S::FWD_BYTES);
    const int per_cu = 2 * S::FWD_BYTES <= 160 * 1024 ? 2 : 1;
    int chunks = (256 * per_cu) / (a.heads * a.L);
    if (chunks < 1) chunks = 1;
    if (chunks > a.nwin) chunks = a.nwin;
    hipLaunchKernelGGL((attn2_fwd_kernel<T, D, LFS>), dim3(chunks, a.heads, a.L), dim3(NTH), S::FWD_BYTES, st, a);
    FW_LAUNCH_RET();
}
template <typename T, int D, int LFS, bool DUAL>
int bwd2_launch_d(const AttnArgs& a, hipStream_t st) {
    using S = Smem2<T, D, LFS, DUAL>;
    FW_SET_LDS_ONCE((attn2_bwd_kernel<T, D, LFS, DUAL>), S::BWD_BYTES);
    const int per_cu = 2 * S::BWD_BYTES <= 160 * 1024 ? 2 : 1;
    int chunks = (256 * per_cu) / (a.heads * a.L);
    if (chunks < 1) chunks = 1;
    if (chunks > a.nwin) chunks = a.nwin;
    hipLaunchKernelGGL((attn2_bwd_kernel<T, D, LFS, DUAL>), dim3(chunks, a.heads, a.L), dim3(NTH), S::BWD_BYTES, st, a);
    FW_LAUNCH_RET();
}
template <typename T, int D, int LFS>
int bwd2_launch(const AttnArgs& a, hipStream_t st) {
    static const int dual = getenv("FW_ATTN_DUAL") ? atoi(getenv("FW_ATTN_DUAL")) : 1;
    if constexpr (LFS == 2 && sizeof(T) == 2) { if (!dual) return bwd2_launch_d<T, D, LFS, false>(a, st); }
    return bwd2_launch_d<T, D, LFS, true>(a, st);
}

template <typename T, int D, int NKT, int LFS>
int fwd_launch(const AttnArgs& a, hipStream_t st) {
    using S = Smem<T, D, NKT, LFS>;
    FW_SET_LDS_ONCE((attn_fwd_kernel<T, D, NKT, LFS>), S::FWD_BYTES);
    const int items = a.nwin * a.L * a.heads;
    hipLaunchKernelGGL((attn_fwd_kernel<T, D, NKT, LFS>), dim3(items < 4096 ? items : 4096), dim3(NTH), S::FWD_BYTES, st, a);
    FW_LAUNCH_RET();
}
template <typename T, int D, int NKT, int LFS>
int bwd_launch(const AttnArgs& a, hipStream_t st) {
    using S = SmemB<T, D, NKT, LFS>;
    FW_SET_LDS_ONCE((attn_bwd_kernel<T, D, NKT, LFS>), S::BYTES);
    hipLaunchKernelGGL((attn_bwd_kernel<T, D, NKT, LFS>), dim3(a.chunks, a.heads, a.L), dim3(NTH), S::BYTES, st, a);
    FW_LAUNCH_RET();
}

template <typename T>
int dispatch(bool bwd, int D, int nkt, int lfs, const AttnArgs& a, hipStream_t st) {
    static const int v2 = getenv("FW_ATTN_V2") ? atoi(getenv("FW_ATTN_V2")) : 1;              // 0: the v1 kernels for one key tile too
#define FW_ATT2(DD, FF)                                                                        \
    if (v2 && D == DD && nkt == 1 && a.mode == 0 && lfs == FF)                                 \
        return bwd ? bwd2_launch<T, DD, FF>(a, st) : fwd2_launch<T, DD, FF>(a, st);
    FW_ATT2(56, 0) FW_ATT2(56, 1) FW_ATT2(56, 2) FW_ATT2(28, 0) FW_ATT2(64, 0)
#undef FW_ATT2
    if (v2 && D == 28 && nkt == 2 && a.mode == 1 && a.L == 3 && lfs == 0) return x2_launch<T, 28>(bwd, a, st);
#define FW_ATT(DD, KK, FF)                                                                     \
    if (D == DD && nkt == KK && lfs == FF)                                                     \
        return bwd ? bwd_launch<T, DD, KK, FF>(a, st) : fwd_launch<T, DD, KK, FF>(a, st);
    FW_ATT(56, 1, 0) FW_ATT(56, 1, 1) FW_ATT(56, 1, 2) FW_ATT(28, 1, 0) FW_ATT(28, 2, 0) FW_ATT(64, 1, 0)
#undef FW_ATT
    return -1000;      // unsupported (head_dim, key tiles, lfs) combination
}

}  // namespace

extern "C" int fw_attn_lfs_table_elems() { return OFF_END; }

// dtype: 0 f32 / 1 bf16.  head_dim D in {56, 28} (Uformer) or 64 (ViT encoder: one 64-token "window" per image, zero bias table).  nkt: key tiles of 64 (2 = inter-band, L = 3).
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
                           int shift, float scale, int dq_pad, void* stream) {
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
    FW_CHECK_ARG(dq_pad == 0 || (dq_pad > 0 && dq_pad < 8 && (heads * D + dq_pad) % 8 == 0));
    FW_CHECK_ARG(dq_pad == 0 || ((dq_pad * sz) % (((D * sz) % 16 == 0) ? 16 : 8) == 0 && (D + dq_pad) * sz <= ((D * sz + 63) / 64) * 64));
    a.dq_pad = dq_pad;
    static const int bwd_wgs = getenv("FW_ATTN_BWD_WGS") ? atoi(getenv("FW_ATTN_BWD_WGS")) : 1024;
    int chunks = bwd_wgs / (heads * L);          // 4-wave workgroups, 2 per CU: about two rounds of the chip
    if (chunks < 1) chunks = 1;
    if (chunks > a.nwin) chunks = a.nwin;
    a.chunks = chunks;
    hipStream_t st = (hipStream_t)stream;
    return dtype == FW_DT_BF16 ? dispatch<bf16raw>(true, D, nkt, lfs, a, st) : dispatch<float>(true, D, nkt, lfs, a, st);
}
