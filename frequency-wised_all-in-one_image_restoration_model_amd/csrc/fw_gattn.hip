// Global multi-head attention of the ViT encoder plug-in, forward + backward (BASELINE configs[4]: N = 256 tokens at 256x256).
//
// Replaces, per (image, head):  net/encoder_ViT.py:76-98 (`Attention.forward`)
//     dots = q k^T * scale;  attn = softmax(dots);
//     [attn += sum_i lamb[i] * band_i(attn)        the learned band re-weighting of :85-92, frequency_decompose_type != 'none']
//     attn = dropout(attn);  out = attn v
// q | k | v are the three column blocks of the `to_qkv` output ('b n (h d) -> b h n d', head h at column h * 64).
//
// MI355X design.  head_dim is 64 and N = 64 * NT with NT in {1, 4}: one workgroup of 4 waves owns one 64-query tile of one
// (image, head) and sees ALL N keys at once -- the 64 x N score block lives in registers (wave w: queries 16w..16w+15, all keys;
// 4 * NT MFMA tiles), so the softmax is exact (no online rescaling) and costs two shuffles per row.  Every contraction is an MFMA
// product of LDS tiles (fw_common.h): scores are formed TRANSPOSED (S^T = K Q^T: keys along the registers, queries along the
// lanes), so that the C/D layout's 4-consecutive-rows-per-lane is written back to LDS with one 8/16-byte store as the
// k-contiguous operand of the next product, and V / dO / Q / K are read along their token axis with ds_read_b64_tr_b16.
// Dropout masks are counter-based (fw_common.h: fw_keep): the backward pass re-derives them, nothing is stored.
// The band re-weighting (N = 64 only: the reference sizes its masks dim_head x dim_head, encoder_ViT.py:56,60) is a full 64x64
// 2-D DFT on the f32 MFMA (exact f32 products): A' = A + Re F^-1( W . F(A) ), W[u][v] = lamb[band(u, v)], with the cos / sin
// panels read as ready-made fragments from L2; it is self-adjoint (real radial W), so the backward pass runs the same routine on
// the incoming gradient and gets d(lamb) from the two spectra.
// Backward (flash style, deterministic, no atomics on activations): workgroup (image, head, tile t) first acts for QUERY tile t
// (loops over the key tiles: dQ), then for KEY tile t (loops over the query tiles: dK, dV), recomputing P from the saved
// log-sum-exp; with NT = 1 both roles share one pass.
#include "fw_common.h"

namespace {

struct GAttnArgs {
    const char* q; const char* k; const char* v; long ld;      // T; row = token (b * N + n), head h at column h * 64; ld in elements
    char* out; long ldo;                                        // fwd: O [B*N][heads*64]
    float* lse;                                                 // [B][heads][N]
    int B, heads, N;
    float scale;
    const unsigned* seed; unsigned site; unsigned thresh; float inv_keep;      // dropout on the attention map (thresh == 0: off)
    const float* lamb; int nb; int lamb_batch;                  // lamb [nb][lamb_batch (1 | B)][heads]
    const unsigned char* bandidx;                               // [64][64]: band of spectrum bin (u, v), un-shifted coordinates
    const float* panels;                                        // f32 cos [64][64] then sin [64][64] of 2 pi u i / 64
    // backward
    const char* o;                                              // forward output (dvec kernel only)
    const char* dout; long lddo;
    float* dvec;                                                // [B][heads][N] rowsum(dO . O)
    char* dq; char* dk; char* dv; long ldd;
    float* dlamb;                                               // same layout as lamb, accumulated with atomics
};

constexpr int NTH = 256;
FW_DEV int wave_id() { return threadIdx.x >> 6; }
FW_DEV void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
FW_DEV float col_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
FW_DEV float col_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

template <typename T> struct GG {
    static constexpr int SZ = TT<T>::SZ;
    static constexpr int LDR = 64 * SZ + 16;          // byte stride of a [rows][64] tile of T (16 B pad: conflict-free fragment reads)
    static constexpr int KC = SZ;                     // 64-byte k-chunks over 64 elements
    static constexpr int GR = 4 * SZ;                 // 16-byte granules per row
    static constexpr int TILE = 64 * LDR;
};
constexpr int LDF = 64 * 4 + 16;                      // f32 [64][64] tile of the spectral filter
constexpr int SLOT = 64 * LDF;

// rows x 64 elements, global (row stride ldb bytes) -> LDS tile; whole workgroup
template <typename T, int ROWS> FW_DEV void load_tile(char* tile, const char* g, long ldb) {
    constexpr int GR = GG<T>::GR, LDR = GG<T>::LDR;
#pragma unroll
    for (int i = 0; i < ROWS * GR / NTH; ++i) {
        const int idx = threadIdx.x + i * NTH, r = idx / GR, s = idx % GR;
        *reinterpret_cast<uint4*>(tile + r * LDR + s * 16) = *reinterpret_cast<const uint4*>(g + r * ldb + s * 16);
    }
}
// 16 rows of an LDS tile -> global; ONE wave
template <typename T> FW_DEV void store_rows16(const char* tile, char* g, long ldb) {
    constexpr int GR = GG<T>::GR, LDR = GG<T>::LDR;
    const int l = lane_id();
#pragma unroll
    for (int i = 0; i < 16 * GR / 64; ++i) {
        const int idx = l + i * 64, r = idx / GR, s = idx % GR;
        *reinterpret_cast<uint4*>(g + r * ldb + s * 16) = *reinterpret_cast<const uint4*>(tile + r * LDR + s * 16);
    }
}
FW_DEV uint4 frag_g(const float* P, int row0, int c) {        // A-operand fragment of a global f32 [64][64] panel
    const int l = lane_id();
    return *reinterpret_cast<const uint4*>(P + (row0 + (l & 15)) * 64 + c * 16 + ((l >> 4) << 2));
}
FW_DEV f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ---- spectral re-weighting on a 64x64 f32 matrix held in strip layout ---------------------------------------------------------
// in[mt][r] = A[16 mt + 4 (l >> 4) + r][16 w + (l & 15)];  out = Re IDFT2( wv . DFT2(A) ) in the same layout;  wv = W at the lane's
// (u = 16 mt + 4 (l >> 4) + r, v = 16 w + (l & 15)); xr / xi receive the (transposed-index) spectrum before weighting.
// arena: 4 slots of SLOT bytes no wave still uses on entry (the entry barrier makes a second call safe); 3 more barriers inside.
FW_DEV void spectral_filter(const f32x4 (&in)[4], f32x4 (&out)[4], const f32x4 (&wv)[4], f32x4 (&xr)[4], f32x4 (&xi)[4], char* arena,
                            const float* Cg, const float* Sg) {
    const int w = wave_id();
    char* As = arena;                 // [kappa][rho], later Yr / Zr
    char* Tr = arena + SLOT;
    char* Ti = arena + 2 * SLOT;
    char* Yi = arena + 3 * SLOT;
    char* Yr = As;
    char* mine = As + 16 * w * LDF;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) store_acc_T<float>(mine, LDF, 16 * mt, 0, in[mt]);
    wave_fence();
    f32x4 p1[4], p2[4], p3[4], p4[4];
    // G1: T[kappa][v] = sum_rho A[rho][kappa] F[rho][v], F = C - iS;  acc(m = v, n = kappa)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { p1[mt] = zero4(); p2[mt] = zero4(); }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 bf = frag_kc(mine, LDF, 0, c);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            mma_chunk<float>(p1[mt], frag_g(Cg, 16 * mt, c), bf);
            mma_chunk<float>(p2[mt], frag_g(Sg, 16 * mt, c), bf);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        store_acc_T<float>(Tr + 16 * w * LDF, LDF, 16 * mt, 0, p1[mt]);          // Ts[kappa][v], v contiguous
        store_acc_T<float>(Ti + 16 * w * LDF, LDF, 16 * mt, 0, -p2[mt]);
    }
    __syncthreads();
    // G2: X[u][v] = sum_kappa F[u][kappa] T[kappa][v];  acc(m = u, n = v own strip); B operand = Ts read k-major
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { p1[mt] = zero4(); p2[mt] = zero4(); p3[mt] = zero4(); p4[mt] = zero4(); }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 br = frag_km<float>(Tr, LDF, 16 * w, c), bi = frag_km<float>(Ti, LDF, 16 * w, c);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const uint4 cf = frag_g(Cg, 16 * mt, c), sf = frag_g(Sg, 16 * mt, c);
            mma_chunk<float>(p1[mt], cf, br); mma_chunk<float>(p2[mt], sf, bi);
            mma_chunk<float>(p3[mt], cf, bi); mma_chunk<float>(p4[mt], sf, br);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        xr[mt] = p1[mt] + p2[mt];                                               // (C - iS)(Tr + iTi)
        xi[mt] = p3[mt] - p4[mt];
        store_acc_T<float>(Yr + 16 * w * LDF, LDF, 16 * mt, 0, xr[mt] * wv[mt]); // Ys[v][u], u contiguous, rows v = own strip
        store_acc_T<float>(Yi + 16 * w * LDF, LDF, 16 * mt, 0, xi[mt] * wv[mt]);
    }
    wave_fence();
    // G3: Z[kappa][v] = sum_u conj(F)[kappa][u] Y[u][v];  acc(m = kappa, n = v own strip)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { p1[mt] = zero4(); p2[mt] = zero4(); p3[mt] = zero4(); p4[mt] = zero4(); }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 br = frag_kc(Yr + 16 * w * LDF, LDF, 0, c), bi = frag_kc(Yi + 16 * w * LDF, LDF, 0, c);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const uint4 cf = frag_g(Cg, 16 * mt, c), sf = frag_g(Sg, 16 * mt, c);
            mma_chunk<float>(p1[mt], cf, br); mma_chunk<float>(p2[mt], sf, bi);
            mma_chunk<float>(p3[mt], cf, bi); mma_chunk<float>(p4[mt], sf, br);
        }
    }
    wave_fence();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {                                            // (C + iS)(Yr + iYi) -> Zs[v][kappa], kappa contiguous
        store_acc_T<float>(Yr + 16 * w * LDF, LDF, 16 * mt, 0, p1[mt] - p2[mt]);
        store_acc_T<float>(Yi + 16 * w * LDF, LDF, 16 * mt, 0, p3[mt] + p4[mt]);
    }
    __syncthreads();
    // G4: out[rho][kappa] = Re sum_v Z[kappa][v] conj(F)[v][rho] / 4096;  acc(m = rho, n = kappa own strip); B = Zs read k-major
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { p1[mt] = zero4(); p2[mt] = zero4(); }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 br = frag_km<float>(Yr, LDF, 16 * w, c), bi = frag_km<float>(Yi, LDF, 16 * w, c);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            mma_chunk<float>(p1[mt], frag_g(Cg, 16 * mt, c), br);
            mma_chunk<float>(p2[mt], frag_g(Sg, 16 * mt, c), bi);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) out[mt] = (p1[mt] - p2[mt]) * (1.0f / 4096.0f);
}

// W at the lane's 16 spectrum bins
FW_DEV void lamb_weights(const GAttnArgs& a, int b, int h, f32x4 (&wv)[4]) {
    const int l = lane_id(), v = 16 * wave_id() + (l & 15);
    const float* lam = a.lamb + (long)(a.lamb_batch > 1 ? b : 0) * a.heads + h;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = 16 * mt + 4 * (l >> 4) + r;
            wv[mt][r] = lam[(long)a.bandidx[u * 64 + v] * a.lamb_batch * a.heads];
        }
}

// ================================================================================================================ forward
template <typename T, int NT, bool LAMB>
__global__ __launch_bounds__(NTH) void gattn_fwd_kernel(GAttnArgs a) {
    using G = GG<T>;
    constexpr int SZ = G::SZ, LDR = G::LDR, KC = G::KC, N = 64 * NT, MT = N / 16;
    constexpr int LDP = N * SZ + 16, JC = N * SZ / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Ks = Qs + 64 * LDR;
    char* Vs = Ks + N * LDR;
    char* arena = Vs + N * LDR;
    const int w = wave_id(), l = lane_id();
    int item = blockIdx.x;
    const int qt = item % NT; item /= NT;
    const int h = item % a.heads, b = item / a.heads;
    const long ldb = a.ld * SZ;
    load_tile<T, 64>(Qs, a.q + ((long)(b * N + qt * 64) * a.ld + h * 64) * SZ, ldb);
    load_tile<T, N>(Ks, a.k + ((long)b * N * a.ld + h * 64) * SZ, ldb);
    load_tile<T, N>(Vs, a.v + ((long)b * N * a.ld + h * 64) * SZ, ldb);
    __syncthreads();
    // S^T strip: s[mt][r] = score(query 16 w + (l & 15), key 16 mt + 4 (l >> 4) + r)
    f32x4 s[MT];
    {
        uint4 qf[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) qf[c] = frag_kc(Qs, LDR, 16 * w, c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            s[mt] = zero4();
#pragma unroll
            for (int c = 0; c < KC; ++c) mma_chunk<T>(s[mt], frag_kc(Ks, LDR, 16 * mt, c), qf[c]);
        }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[mt][r] *= a.scale; mx = fmaxf(mx, s[mt][r]); }
    mx = col_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[mt][r] = __expf(s[mt][r] - mx); sum += s[mt][r]; }
    sum = col_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) s[mt] *= inv;
    const int iq = qt * 64 + 16 * w + (l & 15);                        // the lane's query
    if ((l >> 4) == 0) a.lse[(long)(b * a.heads + h) * N + iq] = mx + __logf(sum);
    if constexpr (LAMB) {
        f32x4 wv[4], fo[4], xr[4], xi[4];
        lamb_weights(a, b, h, wv);
        spectral_filter(s, fo, wv, xr, xi, arena, a.panels, a.panels + 4096);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s[mt] += fo[mt];
    }
    if (a.thresh) {
        const unsigned key = fw_site_key(a.seed[0], a.site);
        const unsigned long long base = ((unsigned long long)(b * a.heads + h) * N + iq) * N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s[mt][r] = fw_keep(key, base + 16 * mt + 4 * (l >> 4) + r, a.thresh) ? s[mt][r] * a.inv_keep : 0.f;
    }
    __syncthreads();                                                   // every wave is done with K: its space takes the P strips
    char* Ps = Ks + w * 16 * LDP;                                      // [16 queries][N keys] of T
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) store_acc_T<T>(Ps, LDP, 16 * mt, 0, s[mt]);
    wave_fence();
    // O^T[d][i] = sum_j V[j][d] P[i][j]:  A = V read along its token axis, B = the P strip
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = zero4();
#pragma unroll
    for (int c = 0; c < JC; ++c) {
        const uint4 pf = frag_kc(Ps, LDP, 0, c);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) mma_chunk<T>(o[dt], frag_km<T>(Vs, LDR, 16 * dt, c), pf);
    }
    char* Os = Qs + 16 * w * LDR;                                      // rows of Q only this wave ever read
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store_acc_T<T>(Os, LDR, 16 * dt, 0, o[dt]);
    wave_fence();
    store_rows16<T>(Os, a.out + ((long)(b * N + qt * 64 + 16 * w) * a.ldo + h * 64) * SZ, a.ldo * SZ);
}

// ================================================================================================================ backward
// dvec[b][h][i] = sum_d dO[i][d] O[i][d]
template <typename T>
__global__ void gattn_dvec_kernel(GAttnArgs a) {
    const long n = (long)a.B * a.N * a.heads;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i % a.heads);
        const long row = i / a.heads;
        const T* o = reinterpret_cast<const T*>(a.o) + row * a.ldo + h * 64;
        const T* d = reinterpret_cast<const T*>(a.dout) + row * a.lddo + h * 64;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 64 / TT<T>::E16; ++g) {
            float fo[8], fd[8];
            unpack16<T>(*reinterpret_cast<const uint4*>(o + g * TT<T>::E16), fo);
            unpack16<T>(*reinterpret_cast<const uint4*>(d + g * TT<T>::E16), fd);
#pragma unroll
            for (int e = 0; e < TT<T>::E16; ++e) s += fo[e] * fd[e];
        }
        const int bb = (int)(row / a.N), t = (int)(row % a.N);
        a.dvec[((long)bb * a.heads + h) * a.N + t] = s;
    }
}

// One (query tile, key tile) pair: from the LDS tiles Qs / dOs (queries) and Ks / Vs (keys) to the strips
//   p2[mt][r] = P''^T  (what multiplies V: after re-weighting and dropout)      ds[mt][r] = scale * dS^T
// rows = key 16 mt + 4 (l >> 4) + r of the key tile, column = query 16 w + (l & 15) of the query tile.
template <typename T, int NT, bool LAMB>
FW_DEV void pair_grads(const GAttnArgs& a, int b, int h, int qi, int kj, const char* Qs, const char* dOs, const char* Ks, const char* Vs,
                       char* arena, f32x4 (&p2)[4], f32x4 (&ds)[4]) {
    using G = GG<T>;
    constexpr int LDR = G::LDR, KC = G::KC, N = 64 * NT;
    const int w = wave_id(), l = lane_id();
    const int iq = qi * 64 + 16 * w + (l & 15);
    const long rowid = (long)(b * a.heads + h) * N + iq;
    const float lse = a.lse[rowid];
    f32x4 p[4], dp[4];
    {
        uint4 qf[KC], df[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) { qf[c] = frag_kc(Qs, LDR, 16 * w, c); df[c] = frag_kc(dOs, LDR, 16 * w, c); }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            p[mt] = zero4(); dp[mt] = zero4();
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                mma_chunk<T>(p[mt], frag_kc(Ks, LDR, 16 * mt, c), qf[c]);
                mma_chunk<T>(dp[mt], frag_kc(Vs, LDR, 16 * mt, c), df[c]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[mt][r] = __expf(p[mt][r] * a.scale - lse);
    f32x4 wv[4], xr[4], xi[4];
    if constexpr (LAMB) {
        f32x4 fo[4];
        lamb_weights(a, b, h, wv);
        spectral_filter(p, fo, wv, xr, xi, arena, a.panels, a.panels + 4096);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) p2[mt] = p[mt] + fo[mt];
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) p2[mt] = p[mt];
    }
    if (a.thresh) {
        const unsigned key = fw_site_key(a.seed[0], a.site);
        const unsigned long long base = (unsigned long long)rowid * N + kj * 64;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool keep = fw_keep(key, base + 16 * mt + 4 * (l >> 4) + r, a.thresh);
                p2[mt][r] = keep ? p2[mt][r] * a.inv_keep : 0.f;
                dp[mt][r] = keep ? dp[mt][r] * a.inv_keep : 0.f;
            }
    }
    float dsum;
    if constexpr (LAMB) {
        // dP = dP' + filter(dP') (self-adjoint);  d lamb[band] = sum over the band's bins of Re( X_P conj(X_dP') ) / 4096
        f32x4 fo[4], yr[4], yi[4];
        spectral_filter(dp, fo, wv, yr, yi, arena, a.panels, a.panels + 4096);
        const int v = 16 * w + (l & 15);
        for (int band = 0; band < a.nb; ++band) {
            float acc = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (a.bandidx[(16 * mt + 4 * (l >> 4) + r) * 64 + v] == band) acc += xr[mt][r] * yr[mt][r] + xi[mt][r] * yi[mt][r];
            acc = wave_sum(acc);
            if (l == 0) atomicAdd(a.dlamb + ((long)band * a.lamb_batch + (a.lamb_batch > 1 ? b : 0)) * a.heads + h, acc * (1.0f / 4096.0f));
        }
        dsum = 0.f;                                                   // the filter mixes rows: D_i = sum_j P dP has to be formed here
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            dp[mt] += fo[mt];
#pragma unroll
            for (int r = 0; r < 4; ++r) dsum += p[mt][r] * dp[mt][r];
        }
        dsum = col_sum(dsum);
    } else {
        dsum = a.dvec[rowid];
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[mt][r] = p[mt][r] * (dp[mt][r] - dsum) * a.scale;
}

template <typename T, int NT, bool LAMB>
__global__ __launch_bounds__(NTH) void gattn_bwd_kernel(GAttnArgs a) {
    using G = GG<T>;
    constexpr int SZ = G::SZ, LDR = G::LDR, KC = G::KC, N = 64 * NT, TILE = G::TILE;
    static_assert(!LAMB || NT == 1, "the band re-weighting is defined for N = 64 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* dOs = Qs + TILE;
    char* Ks = dOs + TILE;
    char* Vs = Ks + TILE;
    char* arena = Vs + TILE;                                           // LAMB: 4 f32 slots; the P'' / dS strips then live in slots 2 / 1
    char* Ps = LAMB ? arena + 2 * SLOT : Vs + TILE;
    char* dSs = LAMB ? arena + SLOT : Ps + TILE;
    const int w = wave_id();
    int item = blockIdx.x;
    const int t = item % NT; item /= NT;
    const int h = item % a.heads, b = item / a.heads;
    const long ldb = a.ld * SZ, lddob = a.lddo * SZ, lddb = a.ldd * SZ;
    const long col = (long)h * 64 * SZ;
    auto rows = [&](const char* base, long ldbytes, int tile) { return base + (long)(b * N + tile * 64) * ldbytes + col; };
    f32x4 p2[4], ds[4], dq[4], dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dq[i] = zero4(); dk[i] = zero4(); dv[i] = zero4(); }

    // ---- role Q: queries of tile t against every key tile (NT == 1: the one pair also feeds dK / dV)
    load_tile<T, 64>(Qs, rows(a.q, ldb, t), ldb);
    load_tile<T, 64>(dOs, rows(a.dout, lddob, t), lddob);
    for (int kj = 0; kj < NT; ++kj) {
        load_tile<T, 64>(Ks, rows(a.k, ldb, kj), ldb);
        load_tile<T, 64>(Vs, rows(a.v, ldb, kj), ldb);
        __syncthreads();
        pair_grads<T, NT, LAMB>(a, b, h, t, kj, Qs, dOs, Ks, Vs, arena, p2, ds);
        char* myS = dSs + 16 * w * LDR;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) store_acc_T<T>(myS, LDR, 16 * mt, 0, ds[mt]);             // dSs[i][j], j contiguous
        if constexpr (NT == 1) {
            char* myP = Ps + 16 * w * LDR;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) store_acc_T<T>(myP, LDR, 16 * mt, 0, p2[mt]);
        }
        wave_fence();
        // dQ^T[d][i] += sum_j K[j][d] dS[i][j]
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const uint4 bf = frag_kc(myS, LDR, 0, c);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) mma_chunk<T>(dq[dt], frag_km<T>(Ks, LDR, 16 * dt, c), bf);
        }
        if constexpr (NT == 1) {
            __syncthreads();
            // dV^T[d][j] = sum_i dO[i][d] P''[i][j];  dK^T[d][j] = sum_i Q[i][d] dS[i][j]   (wave w: keys 16 w ..)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const uint4 pf = frag_km<T>(Ps, LDR, 16 * w, c), sf = frag_km<T>(dSs, LDR, 16 * w, c);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    mma_chunk<T>(dv[dt], frag_km<T>(dOs, LDR, 16 * dt, c), pf);
                    mma_chunk<T>(dk[dt], frag_km<T>(Qs, LDR, 16 * dt, c), sf);
                }
            }
        }
        __syncthreads();
    }
    {   // every wave has left the tiles: stage through the Q / K / V rows of the own strip
        char* st = Qs + 16 * w * LDR;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store_acc_T<T>(st, LDR, 16 * dt, 0, dq[dt]);
        if constexpr (NT == 1) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store_acc_T<T>(Ks + 16 * w * LDR, LDR, 16 * dt, 0, dk[dt]);
                store_acc_T<T>(Vs + 16 * w * LDR, LDR, 16 * dt, 0, dv[dt]);
            }
        }
        wave_fence();
        store_rows16<T>(st, const_cast<char*>(rows(a.dq, lddb, t)) + (long)16 * w * lddb, lddb);
        if constexpr (NT == 1) {
            store_rows16<T>(Ks + 16 * w * LDR, const_cast<char*>(rows(a.dk, lddb, t)) + (long)16 * w * lddb, lddb);
            store_rows16<T>(Vs + 16 * w * LDR, const_cast<char*>(rows(a.dv, lddb, t)) + (long)16 * w * lddb, lddb);
        }
    }
    if constexpr (NT > 1) {
        // ---- role K: keys of tile t against every query tile
        __syncthreads();
        load_tile<T, 64>(Ks, rows(a.k, ldb, t), ldb);
        load_tile<T, 64>(Vs, rows(a.v, ldb, t), ldb);
        for (int qi = 0; qi < NT; ++qi) {
            load_tile<T, 64>(Qs, rows(a.q, ldb, qi), ldb);
            load_tile<T, 64>(dOs, rows(a.dout, lddob, qi), lddob);
            __syncthreads();
            pair_grads<T, NT, LAMB>(a, b, h, qi, t, Qs, dOs, Ks, Vs, arena, p2, ds);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                store_acc_T<T>(Ps + 16 * w * LDR, LDR, 16 * mt, 0, p2[mt]);
                store_acc_T<T>(dSs + 16 * w * LDR, LDR, 16 * mt, 0, ds[mt]);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const uint4 pf = frag_km<T>(Ps, LDR, 16 * w, c), sf = frag_km<T>(dSs, LDR, 16 * w, c);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    mma_chunk<T>(dv[dt], frag_km<T>(dOs, LDR, 16 * dt, c), pf);
                    mma_chunk<T>(dk[dt], frag_km<T>(Qs, LDR, 16 * dt, c), sf);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            store_acc_T<T>(Ks + 16 * w * LDR, LDR, 16 * dt, 0, dk[dt]);
            store_acc_T<T>(Vs + 16 * w * LDR, LDR, 16 * dt, 0, dv[dt]);
        }
        wave_fence();
        store_rows16<T>(Ks + 16 * w * LDR, const_cast<char*>(rows(a.dk, lddb, t)) + (long)16 * w * lddb, lddb);
        store_rows16<T>(Vs + 16 * w * LDR, const_cast<char*>(rows(a.dv, lddb, t)) + (long)16 * w * lddb, lddb);
    }
}

template <typename T> static size_t fwd_lds(int NT, bool lamb) {
    return (size_t)(64 + 2 * 64 * NT) * GG<T>::LDR + (lamb ? 4 * SLOT : 0);
}
template <typename T> static size_t bwd_lds(bool lamb) { return (size_t)4 * GG<T>::TILE + (lamb ? 4 * SLOT : 2 * GG<T>::TILE); }

template <typename T, int NT, bool LAMB> static void launch_fwd(const GAttnArgs& a, hipStream_t st) {
    const size_t lds = fwd_lds<T>(NT, LAMB);
    FW_SET_LDS_ONCE((gattn_fwd_kernel<T, NT, LAMB>), lds);
    hipLaunchKernelGGL((gattn_fwd_kernel<T, NT, LAMB>), dim3(a.B * a.heads * NT), dim3(NTH), lds, st, a);
}
template <typename T, int NT, bool LAMB> static void launch_bwd(const GAttnArgs& a, hipStream_t st) {
    if (!LAMB) {
        const long n = (long)a.B * a.N * a.heads;
        hipLaunchKernelGGL((gattn_dvec_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    }
    const size_t lds = bwd_lds<T>(LAMB);
    FW_SET_LDS_ONCE((gattn_bwd_kernel<T, NT, LAMB>), lds);
    hipLaunchKernelGGL((gattn_bwd_kernel<T, NT, LAMB>), dim3(a.B * a.heads * NT), dim3(NTH), lds, st, a);
}
template <typename T> static int dispatch(const GAttnArgs& a, bool bwd, hipStream_t st) {
    const bool lamb = a.lamb != nullptr;
    if (a.N == 64) {
        if (lamb) bwd ? launch_bwd<T, 1, true>(a, st) : launch_fwd<T, 1, true>(a, st);
        else bwd ? launch_bwd<T, 1, false>(a, st) : launch_fwd<T, 1, false>(a, st);
    } else {
        bwd ? launch_bwd<T, 4, false>(a, st) : launch_fwd<T, 4, false>(a, st);
    }
    FW_LAUNCH_RET();
}
static bool common_ok(const GAttnArgs& a, int dtype) {
    const int sz = dtype == FW_DT_BF16 ? 2 : 4;
    if (!(a.q && a.k && a.v && a.lse && a.B > 0 && a.heads > 0 && (a.N == 64 || a.N == 256))) return false;
    if ((a.ld * sz) % 16 || ((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) % 16) return false;
    if (a.thresh && !a.seed) return false;
    if (a.lamb && !(a.N == 64 && a.bandidx && a.panels && a.nb >= 1 && a.nb <= 16 && (a.lamb_batch == 1 || a.lamb_batch == a.B))) return false;
    return true;
}
}  // namespace

extern "C" int fw_gattn_fwd(int dtype, const void* q, const void* k, const void* v, long ld, void* out, long ldo, float* lse, int B, int heads,
                            int N, float scale, const void* seed, int site, float drop_p, const float* lamb, int nb, int lamb_batch,
                            const void* bandidx, const float* panels, void* stream) {
    GAttnArgs a{};
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld = ld; a.out = (char*)out; a.ldo = ldo; a.lse = lse;
    a.B = B; a.heads = heads; a.N = N; a.scale = scale;
    a.seed = (const unsigned*)seed; a.site = (unsigned)site; a.thresh = drop_p > 0.f ? fw_drop_thresh(drop_p) : 0u;
    a.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    a.lamb = lamb; a.nb = nb; a.lamb_batch = lamb_batch; a.bandidx = (const unsigned char*)bandidx; a.panels = panels;
    const int sz = dtype == FW_DT_BF16 ? 2 : 4;
    FW_CHECK_ARG(dtype == FW_DT_BF16 || dtype == FW_DT_F32);
    FW_CHECK_ARG(common_ok(a, dtype) && out && (ldo * sz) % 16 == 0 && (uintptr_t)out % 16 == 0 && drop_p >= 0.f && drop_p < 1.f);
    return dtype == FW_DT_BF16 ? dispatch<bf16raw>(a, false, (hipStream_t)stream) : dispatch<float>(a, false, (hipStream_t)stream);
}

extern "C" int fw_gattn_bwd(int dtype, const void* q, const void* k, const void* v, long ld, const void* o, long ldo, const void* dout, long lddo,
                            const float* lse, float* dvec, void* dq, void* dk, void* dv, long ldd, int B, int heads, int N, float scale,
                            const void* seed, int site, float drop_p, const float* lamb, float* dlamb, int nb, int lamb_batch,
                            const void* bandidx, const float* panels, void* stream) {
    GAttnArgs a{};
    a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld = ld; a.o = (const char*)o; a.ldo = ldo;
    a.dout = (const char*)dout; a.lddo = lddo; a.lse = const_cast<float*>(lse); a.dvec = dvec;
    a.dq = (char*)dq; a.dk = (char*)dk; a.dv = (char*)dv; a.ldd = ldd;
    a.B = B; a.heads = heads; a.N = N; a.scale = scale;
    a.seed = (const unsigned*)seed; a.site = (unsigned)site; a.thresh = drop_p > 0.f ? fw_drop_thresh(drop_p) : 0u;
    a.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    a.lamb = lamb; a.dlamb = dlamb; a.nb = nb; a.lamb_batch = lamb_batch; a.bandidx = (const unsigned char*)bandidx; a.panels = panels;
    const int sz = dtype == FW_DT_BF16 ? 2 : 4;
    FW_CHECK_ARG(dtype == FW_DT_BF16 || dtype == FW_DT_F32);
    FW_CHECK_ARG(common_ok(a, dtype) && o && dout && dq && dk && dv && (lamb ? dlamb != nullptr : dvec != nullptr));
    FW_CHECK_ARG((ldo * sz) % 16 == 0 && (lddo * sz) % 16 == 0 && (ldd * sz) % 16 == 0 && drop_p >= 0.f && drop_p < 1.f);
    FW_CHECK_ARG(((uintptr_t)o | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0);
    return dtype == FW_DT_BF16 ? dispatch<bf16raw>(a, true, (hipStream_t)stream) : dispatch<float>(a, true, (hipStream_t)stream);
}
