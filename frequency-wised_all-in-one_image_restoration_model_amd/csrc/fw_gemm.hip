// fw_gemm: the one GEMM of the AirNet hot path (K3/K4/K7/K8 of SURVEY.md section 2.2).
//
//   C[m][n] = epilogue( alpha * sum_k X(m,k) * W(n,k) )
//
// Replaces the reference's nn.Linear forward/backward (decoder_Uformer.py:98-125 LinearProjection,
// :294 proj, net/utils/leff.py:100,114 linear1/linear2, encoder_Uformer.py:942 mlp_head) and, with
// the im2col helpers of fw_elem.hip, its k4s2 / k2s2 convolutions (decoder_Uformer.py:414-449).
//
// MI355X design: 256 threads = 4 waves per workgroup, one 128(m) x BN(n) output tile, K walked in
// 128-byte steps (64 bf16 / 32 f32) through a double-buffered, register-staged LDS pipeline with
// one barrier per step.  Both operands are held in LDS k-contiguous ([row][k], 144-byte padded
// rows).  An operand whose reduction index is the SLOW axis in HBM (dX = dY*W, dW = dY^T*X) is
// transposed in registers while it is staged (16-byte loads along the fast axis, v_perm_b32 /
// register renaming, 8- or 16-byte LDS writes), so one MFMA inner loop serves NT, NN and TN.
// The MFMA roles are swapped (A-operand = W rows, B-operand = X rows) so that a lane ends up
// with 4 CONSECUTIVE n for one m: bias / residual / output are 8- and 16-byte vector accesses.
// bf16 uses v_mfma_f32_16x16x32_bf16, f32 uses v_mfma_f32_16x16x4_f32 (exact f32), same code.
#include "fw_common.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

namespace {
// name of the kernel the last fw_gemm call of this thread launched, spelt like rocprofv3's demangled kernel names with spaces
// removed and "unsigned short" written bf16 (measurement aid: bench.py and tools/pmc_summary.py key their per-kernel rows on it)
static thread_local char g_last_kernel[96] = "";
template <typename T> constexpr const char* tname() { return sizeof(T) == 2 ? "bf16" : "float"; }
#define FW_B(x) ((x) ? "true" : "false")
#define FW_KNAME(...) snprintf(g_last_kernel, sizeof g_last_kernel, __VA_ARGS__)

struct GemmArgs {
    const char* X; const char* W; char* C;
    long ldx, ldw, ldc;                 // elements
    int M, N, K;
    int x_op, w_op;                     // 1: GELU applied to the operand while staging
    const float* bias;                  // [N] or null
    int act; float slope;               // 1: leaky relu   2: times gelu'(aux)   3: gelu
    const char* aux; long ldaux;        // T [M][ldaux]
    const float* rowscale; int rows_per_scale;
    const float* residual; long ldr;    // f32 [M][ldr]
    int out_f32;                        // C is float (else T)
    int accumulate;                     // 0 store, 1 atomicAdd (f32 C only)
    int splitk; int kper;               // K range per z-slice (multiple of the K step)
    char* C2; long ldc2;                // optional second output GELU(v), type T
    float* xsum;                        // optional: xsum[m] += sum_k X(m,k) (x_trans only; the bias gradient of a dW GEMM)
    long c_zstride, xsum_zstride;       // > 0: split-K slice z writes its partial tile / sums to C + z*stride (plain stores, no atomics)
    float alpha;
    int staged;                         // >= 0: bf16 epilogue through LDS (tile_epilogue_staged mode)
    int dbg;                            // measurement only (FW_GEMM_BIG_DBG): 1 skip the epilogue's stores, 2 skip the operand loads
};

constexpr int LDS_ROW = 128;            // 128 B of K per row, XOR-swizzled in 16-byte slots (no padding)
// slot swizzle: spreads both the row-strided fragment reads (rows r..r+15, same slot) and the 8-row-strided
// transposed staging writes (rows 8*ib+e) over the 8 slots of a row
FW_DEV int swz(int row) { return ((row ^ (row >> 3)) & 7) << 4; }
constexpr int BM = 128;

// Workgroups are dealt round-robin to the 8 XCDs (8 private L2s): block `lin` of `total` -> position in an order where every
// XCD owns one CONTIGUOUS share (bijective for any total: the first total % 8 XCDs take one block more;
// cdna_hip_programming.md 5.5 T1).  Speed only, never correctness.
FW_DEV unsigned xcd_contiguous(unsigned lin, unsigned total) {
    const unsigned xcd = lin & 7, q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
}

template <typename T> FW_DEV uint4 apply_gelu16(const uint4& v) {
    float f[TT<T>::E16];
    unpack16<T>(v, f);
#pragma unroll
    for (int i = 0; i < TT<T>::E16; ++i) f[i] = gelu_t<T>(f[i]);
    return pack16<T>(f);
}

// ---- epilogue of one lane: 4 consecutive n of row m ------------------------------------------
// Two phases so that the loads of SEVERAL quads are in flight together: epi_fetch issues the operand loads of a quad from
// clamped coordinates under wave-uniform conditions only (no lane-varying branch, nothing consumes the result yet);
// epi_apply does the arithmetic and the stores and is the only part called under the lane's validity test.
// ext carries ONE row-dependent operand of the quad: GELU'(aux) input when act == 2, else the f32 residual (when both are
// given -- not in this model -- the residual is read inside epi_apply).  The bias depends on n only: fetched once per column.
template <typename T>
FW_DEV void epi_fetch(const GemmArgs& a, uint4& ext, int mc, int nc, int z) {
    if (a.act == 2) {
        const T* ap = reinterpret_cast<const T*>(a.aux) + (long)mc * a.ldaux + nc;
        if (sizeof(T) == 4) ext = *reinterpret_cast<const uint4*>(ap);
        else { const uint2 t = *reinterpret_cast<const uint2*>(ap); ext.x = t.x; ext.y = t.y; }
    } else if (a.residual && z == 0) {
        ext = *reinterpret_cast<const uint4*>(a.residual + (long)mc * a.ldr + nc);
    }
}
FW_DEV f32x4 epi_bias(const GemmArgs& a, int nc, int z) {
    return (a.bias && z == 0) ? *reinterpret_cast<const f32x4*>(a.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
}
template <typename T>
FW_DEV void epi_apply(const GemmArgs& a, const f32x4& bias, const uint4& ext, const f32x4& acc, int m, int n0, float rs, int z) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] * a.alpha + bias[r];
    if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = lrelu_f(v[r], a.slope);
    } else if (a.act == 2) {
        float x[4];
        if (sizeof(T) == 4) { x[0] = __uint_as_float(ext.x); x[1] = __uint_as_float(ext.y); x[2] = __uint_as_float(ext.z); x[3] = __uint_as_float(ext.w); }
        else { x[0] = __uint_as_float(ext.x << 16); x[1] = __uint_as_float(ext.x & 0xffff0000u);
               x[2] = __uint_as_float(ext.y << 16); x[3] = __uint_as_float(ext.y & 0xffff0000u); }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_t<T>(x[r]);
    } else if (a.act == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_t<T>(v[r]);
    }
    if (a.rowscale) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
    }
    if (a.residual && z == 0) {
        uint4 rr = ext;
        if (a.act == 2) rr = *reinterpret_cast<const uint4*>(a.residual + (long)m * a.ldr + n0);
        v[0] += __uint_as_float(rr.x); v[1] += __uint_as_float(rr.y); v[2] += __uint_as_float(rr.z); v[3] += __uint_as_float(rr.w);
    }
    if (a.C2) {
        T* c2 = reinterpret_cast<T*>(a.C2) + (long)m * a.ldc2 + n0;
        const float g0 = gelu_t<T>(v[0]), g1 = gelu_t<T>(v[1]), g2 = gelu_t<T>(v[2]), g3 = gelu_t<T>(v[3]);
        if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(c2) = f32x4{g0, g1, g2, g3};
        else *reinterpret_cast<uint2*>(c2) = make_uint2(pack_bf2(g0, g1), pack_bf2(g2, g3));
    }
    if (a.out_f32) {
        float* cp = reinterpret_cast<float*>(a.C) + (long)z * a.c_zstride + (long)m * a.ldc + n0;
        if (a.accumulate && a.c_zstride == 0) {
            if (a.splitk == 1) {                        // every element has exactly one producer in this launch: plain 16-byte read-modify-write
                const f32x4 old = *reinterpret_cast<const f32x4*>(cp);
                *reinterpret_cast<f32x4*>(cp) = f32x4{old[0] + v[0], old[1] + v[1], old[2] + v[2], old[3] + v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
            }
        } else {
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
        }
    } else {
        T* cp = reinterpret_cast<T*>(a.C) + (long)m * a.ldc + n0;
        if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
        else *reinterpret_cast<uint2*>(cp) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    }
}

// ---- operand staging -----------------------------------------------------------------------
// DIRECT: element (i, k) at base + i*ld + k.   ROWS rows x 8 chunks of 16 B.
template <typename T, int ROWS>
struct StageDirect {
    static constexpr int NL = ROWS / 32;
    uint4 r[NL];
    // Branch-free and two-phase: a load under a lane-varying branch (or followed at once by code that touches its destination)
    // makes the wave wait for it before the next load is issued -- a chain of memory latencies per K step.  All loads are issued
    // first, from clamped coordinates; out-of-range data is zeroed by selects afterwards.
    FW_MEM void load(const char* base, long ld, int row0, int rows_total, int kbyte0, int kbytes_end, int op) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const int cid = tid + 256 * t;
            const int row = cid >> 3, kb = kbyte0 + (cid & 7) * 16;
            const int rr = row0 + row < rows_total ? row0 + row : rows_total - 1;
            r[t] = *reinterpret_cast<const uint4*>(base + (long)rr * ld * TT<T>::SZ + (kb < kbytes_end ? kb : 0));
        }
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const int cid = tid + 256 * t;
            const int row = cid >> 3, kb = kbyte0 + (cid & 7) * 16;
            uint4 v = r[t];
            if (op == 1) v = apply_gelu16<T>(v);
            const int valid = row0 + row < rows_total ? kbytes_end - kb : 0;     // K*sizeof(T) need only be a multiple of 4 bytes
            v.x = valid > 0 ? v.x : 0u; v.y = valid > 4 ? v.y : 0u; v.z = valid > 8 ? v.z : 0u; v.w = valid > 12 ? v.w : 0u;
            r[t] = v;
        }
    }
    FW_MEM void store(char* tile) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int t = 0; t < NL; ++t) {
            const int cid = tid + 256 * t;
            const int row = cid >> 3;
            *reinterpret_cast<uint4*>(tile + row * LDS_ROW + (((cid & 7) * 16) ^ swz(row))) = r[t];
        }
    }
};

// TRANS: element (i, k) at base + k*ld + i  (i contiguous).  A thread owns a 4(k) x E(i) block.
template <typename T, int ROWS>
struct StageTrans {
    static constexpr int E = TT<T>::E16;
    static constexpr int IB = ROWS / E;                 // i-blocks per tile
    static constexpr int KT = 128 / TT<T>::SZ;          // k elements per step
    static constexpr int NTHR = IB * (KT / 4);          // active threads (256 or 128)
    uint4 r[4];
    FW_MEM void load(const char* base, long ld, int row0, int rows_total, int k0, int k_end, int op) {
        const int tid = threadIdx.x;
        const int ib = tid % IB, kb = tid / IB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = k0 + kb * 4 + kk;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tid < NTHR && k < k_end && row0 + ib * E < rows_total) {         // nothing below touches v before the loop ends: the 4 loads stay in flight together
                v = *reinterpret_cast<const uint4*>(base + ((long)k * ld + row0 + ib * E) * TT<T>::SZ);
                if (op == 1) v = apply_gelu16<T>(v);
            }
            r[kk] = v;
        }
    }
    FW_MEM void accum(float (&s)[TT<T>::E16]) const {        // column sums of the staged block (bias gradient)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float f[TT<T>::E16];
            unpack16<T>(r[kk], f);
#pragma unroll
            for (int e = 0; e < TT<T>::E16; ++e) s[e] += f[e];
        }
    }
    FW_MEM void store(char* tile) const {
        const int tid = threadIdx.x;
        if (tid >= NTHR) return;
        const int ib = tid % IB, kb = tid / IB;
        if constexpr (sizeof(T) == 4) {
            const int r0 = ib * 4, cb = kb * 16;
            *reinterpret_cast<uint4*>(tile + (r0 + 0) * LDS_ROW + (cb ^ swz(r0 + 0))) = make_uint4(r[0].x, r[1].x, r[2].x, r[3].x);
            *reinterpret_cast<uint4*>(tile + (r0 + 1) * LDS_ROW + (cb ^ swz(r0 + 1))) = make_uint4(r[0].y, r[1].y, r[2].y, r[3].y);
            *reinterpret_cast<uint4*>(tile + (r0 + 2) * LDS_ROW + (cb ^ swz(r0 + 2))) = make_uint4(r[0].z, r[1].z, r[2].z, r[3].z);
            *reinterpret_cast<uint4*>(tile + (r0 + 3) * LDS_ROW + (cb ^ swz(r0 + 3))) = make_uint4(r[0].w, r[1].w, r[2].w, r[3].w);
        } else {
            const int r0 = ib * 8, cb = kb * 8;
            const unsigned w0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
            const unsigned w1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
            const unsigned w2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
            const unsigned w3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // element e = 2d (low halves) and e = 2d+1 (high halves) of the four k-rows
                uint2 lo = make_uint2(__builtin_amdgcn_perm(w1[d], w0[d], 0x05040100u),
                                      __builtin_amdgcn_perm(w3[d], w2[d], 0x05040100u));
                uint2 hi = make_uint2(__builtin_amdgcn_perm(w1[d], w0[d], 0x07060302u),
                                      __builtin_amdgcn_perm(w3[d], w2[d], 0x07060302u));
                *reinterpret_cast<uint2*>(tile + (r0 + 2 * d) * LDS_ROW + (cb ^ swz(r0 + 2 * d))) = lo;
                *reinterpret_cast<uint2*>(tile + (r0 + 2 * d + 1) * LDS_ROW + (cb ^ swz(r0 + 2 * d + 1))) = hi;
            }
        }
    }
};

// GLDS: direct global -> LDS (global_load_lds_dwordx4), no VGPR round trip and no ds_write.  One wave-instruction fills
// 8 rows x 128 B linearly (LDS address = wave-uniform base + lane*16), so the XOR slot swizzle is applied to the per-lane
// SOURCE address: physical slot p of row r receives logical chunk p ^ sw(r).  Needs whole K steps and no load-op; rows past
// the end are clamped (they only feed masked outputs).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
template <typename T, int ROWS>
FW_DEV void glds_issue(const char* base, long ld, int row0, int rows_total, int kbyte0, char* tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) {
        const int R0 = (wave * (ROWS / 32) + it) * 8;
        const int r = R0 + (lane >> 3), p = lane & 7;
        int gr = row0 + r;
        if (gr >= rows_total) gr = rows_total - 1;
        const char* g = base + (long)gr * ld * TT<T>::SZ + kbyte0 + ((p ^ (swz(r) >> 4)) << 4);
        __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)(tile + R0 * LDS_ROW), 16, 0, 0);
    }
}

template <typename T, int ROWS> struct StageDirectAcc : StageDirect<T, ROWS> {
    FW_MEM void accum(float (&)[TT<T>::E16]) const {}
};
template <typename T, int ROWS, bool TRANS> struct Stage;
template <typename T, int ROWS> struct Stage<T, ROWS, false> : StageDirectAcc<T, ROWS> {
    FW_MEM void fetch(const char* base, long ld, int row0, int rows_total, int k0, int k_end, int op) {
        this->load(base, ld, row0, rows_total, k0 * TT<T>::SZ, k_end * TT<T>::SZ, op);
    }
};
template <typename T, int ROWS> struct Stage<T, ROWS, true> : StageTrans<T, ROWS> {
    FW_MEM void fetch(const char* base, long ld, int row0, int rows_total, int k0, int k_end, int op) {
        this->load(base, ld, row0, rows_total, k0, k_end, op);
    }
};

FW_DEV uint4 frag_sw(const char* tile, int row0, int chunk) {
    const int l = lane_id();
    const int row = row0 + (l & 15);
    return *reinterpret_cast<const uint4*>(tile + row * LDS_ROW + ((chunk * 64 + ((l >> 4) << 4)) ^ swz(row)));
}

template <typename T, int BN, bool XT, bool WT, bool GX, bool GW>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    constexpr int KT = 128 / TT<T>::SZ;                  // k elements per step
    constexpr int WM = (BN == 128) ? 4 : 2;              // m tiles (16) per wave
    constexpr int XBYTES = BM * LDS_ROW, WBYTES = BN * LDS_ROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto xs = [&](int i) -> char* { return smem + i * (XBYTES + WBYTES); };
    auto ws = [&](int i) -> char* { return smem + i * (XBYTES + WBYTES) + XBYTES; };

    const int wave = threadIdx.x >> 6;
    // Workgroups are dealt round-robin to the 8 XCDs (8 private L2s).  With split-K the tiles of one K slice read the same token
    // rows: give every XCD one CONTIGUOUS eighth of the (slice, n tile, m tile) order, so that a slice's operands are fetched into
    // one or two L2s instead of all eight (PMC: 1.9x the algorithmic bytes crossed the fabric with round-robin placement).
    // (Without split-K, tiles sharing X rows are gridDim.x apart and already meet when gridDim.x % 8 == 0.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (gridDim.z > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned w = xcd_contiguous(lin, total);
            bx = (int)(w % gridDim.x);
            by = (int)((w / gridDim.x) % gridDim.y);
            bz = (int)(w / (gridDim.x * gridDim.y));
        } else if (gridDim.z == 1 && gridDim.y > 1) {
            // no split-K: every XCD takes a contiguous eighth of a GROUPED tile order (bands of 8 m tiles, n tiles swept inside a
            // band), so the ~64 tiles an XCD has in flight touch 8 x 8 operand panels that fit its 4 MB L2 instead of re-fetching
            // X once per n tile (PMC: 2.5x the algorithmic bytes crossed the fabric in plain row-major order)
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned w = xcd_contiguous(lin, total);
            const unsigned per_band = 8 * gridDim.y;
            const unsigned band = w / per_band, first = band * 8;
            const unsigned gsz = min(gridDim.x - first, 8u);
            bx = (int)(first + (w % per_band) % gsz);
            by = (int)((w % per_band) / gsz);
        }
    }
    const int m_blk = bx * BM, n_blk = by * BN;
    const int wm0 = (BN == 128) ? (wave & 1) * 64 : wave * 32;
    const int wn0 = (BN == 128) ? (wave >> 1) * 64 : 0;

    const int k_begin = bz * a.kper;
    const int k_end = min(a.K, k_begin + a.kper);
    const int nsteps = (k_end - k_begin + KT - 1) / KT;

    f32x4 acc[4][WM];                                    // [n tile][m tile]: rows = n, cols = m
    zero_acc(acc);

    Stage<T, BM, XT> sx;
    Stage<T, BN, WT> sw;
    float xsum[TT<T>::E16];
#pragma unroll
    for (int e = 0; e < TT<T>::E16; ++e) xsum[e] = 0.f;
    const bool do_xsum = XT && a.xsum != nullptr && by == 0;
    if (nsteps > 0) {
        if constexpr (GX) glds_issue<T, BM>(a.X, a.ldx, m_blk, a.M, k_begin * TT<T>::SZ, xs(0));
        else { sx.fetch(a.X, a.ldx, m_blk, a.M, k_begin, k_end, a.x_op); if (do_xsum) sx.accum(xsum); }
        if constexpr (GW) glds_issue<T, BN>(a.W, a.ldw, n_blk, a.N, k_begin * TT<T>::SZ, ws(0));
        else sw.fetch(a.W, a.ldw, n_blk, a.N, k_begin, k_end, a.w_op);
        if constexpr (!GX) sx.store(xs(0));
        if constexpr (!GW) sw.store(ws(0));
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) {
            if constexpr (GX) glds_issue<T, BM>(a.X, a.ldx, m_blk, a.M, (k_begin + (s + 1) * KT) * TT<T>::SZ, xs(cur ^ 1));
            else { sx.fetch(a.X, a.ldx, m_blk, a.M, k_begin + (s + 1) * KT, k_end, a.x_op); if (do_xsum) sx.accum(xsum); }
            if constexpr (GW) glds_issue<T, BN>(a.W, a.ldw, n_blk, a.N, (k_begin + (s + 1) * KT) * TT<T>::SZ, ws(cur ^ 1));
            else sw.fetch(a.W, a.ldw, n_blk, a.N, k_begin + (s + 1) * KT, k_end, a.w_op);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 af[4], bfr[WM];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = frag_sw(ws(cur), wn0 + 16 * m, c);
#pragma unroll
            for (int n = 0; n < WM; ++n) bfr[n] = frag_sw(xs(cur), wm0 + 16 * n, c);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
        }
        if (s + 1 < nsteps) {
            if constexpr (!GX) sx.store(xs(cur ^ 1));
            if constexpr (!GW) sw.store(ws(cur ^ 1));
        }
        __syncthreads();                                 // also drains the in-flight global_load_lds (vmcnt(0))
    }

    if constexpr (XT) {
        if (a.xsum != nullptr && by == 0) {          // block-uniform: reduce the 16 k-slices on chip, one atomic per row
            constexpr int E = TT<T>::E16, IB = BM / E;
            float* red = reinterpret_cast<float*>(smem);       // the staging tiles are dead after the last barrier
            if (threadIdx.x < BM) red[threadIdx.x] = 0.f;
            __syncthreads();
            if ((int)threadIdx.x < IB * ((128 / TT<T>::SZ) / 4)) {
                const int ib = threadIdx.x % IB;
#pragma unroll
                for (int e = 0; e < E; ++e) atomicAdd(&red[ib * E + e], xsum[e]);
            }
            __syncthreads();
            if (threadIdx.x < BM && m_blk + (int)threadIdx.x < a.M) {
                if (a.xsum_zstride > 0) a.xsum[(long)bz * a.xsum_zstride + m_blk + threadIdx.x] = red[threadIdx.x];
                else atomicAdd(a.xsum + m_blk + threadIdx.x, red[threadIdx.x]);
            }
        }
    }
    // ---- epilogue: lane holds C[m][n0..n0+3], m = col of the MFMA tile, n = rows -------------
    // (an LDS-staged, row-coalesced form of these stores measured ~20 % slower on MI355X: L2 merges the 8-byte pieces)
    const int l = lane_id();
    f32x4 bias4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
        bias4[nt] = epi_bias(a, n0 < a.N ? n0 : 0, bz);
    }
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int m = m_blk + wm0 + mt * 16 + (l & 15);
        const int mc = m < a.M ? m : a.M - 1;
        const float rs = a.rowscale ? a.rowscale[mc / a.rows_per_scale] : 1.0f;
        uint4 ext[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
            epi_fetch<T>(a, ext[nt], mc, n0 < a.N ? n0 : 0, bz);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
            if (m < a.M && n0 < a.N) epi_apply<T>(a, bias4[nt], ext[nt], acc[nt][mt], m, n0, rs, bz);
        }
    }
}

// =====================================================================================================
// Weight-gradient kernel, bf16:  C[m][n] = sum_t X[t][m] W[t][n]  (both operands token-major: dW = dY^T x).
// The tile kernel above transposes both operands in registers while staging them (v_perm + 8-byte LDS writes every K step).
// Here the token-major tiles go to LDS AS THEY ARE -- global_load_lds, no VGPR round trip, no VALU -- and the MFMA fragments,
// which need 8 consecutive TOKENS per lane, are read with gfx950's transposing LDS read (ds_read_b64_tr_b16, two per
// fragment).  LDS image: [64 tokens][128 columns] bf16 = 256-byte rows whose 16-byte chunks are XOR-swizzled by
// f(row) = ((row & 3) << 2) | ((row >> 2) & 3)  (conflict-free for the transposed reads: cdna_hip_programming.md T10 image (b));
// global_load_lds writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE chunk.
// The bias gradient (column sums of X) comes from MFMAs against an all-ones A fragment -- no extra pass over the data.
FW_DEV int swz256(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// one [64 tokens][128 cols] tile; wave w fills token rows 16w .. 16w+15 with 4 wave-instructions of 4 rows each
FW_DEV void glds_issue_km(const char* base, long ld, int col0, int cols_total, int k0, char* tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int R0 = wave * 16 + it * 4;
        const int r = R0 + (lane >> 4), p = lane & 15;
        int col = col0 + ((p ^ swz256(r)) << 3);
        if (col >= cols_total) col = 0;                    // columns past the matrix only feed masked outputs; stay inside the row
        const char* g = base + ((long)(k0 + r) * ld + col) * 2;
        __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)(tile + R0 * 256), 16, 0, 0);
    }
}
// MFMA fragment of operand rows (= tile columns) m0 .. m0+15 over token chunk c (32 tokens): two transposing reads
FW_DEV uint4 frag_tr256(const char* tile, int m0, int chunk) {
    const int l = lane_id();
    const int r = chunk * 32 + ((l >> 4) << 3) + ((l >> 2) & 3);
    const int ch = (m0 >> 3) + ((l & 3) >> 1), sub = (l & 1) << 3;
    const char* p0 = tile + 256 * r + ((ch ^ swz256(r)) << 4) + sub;
    const char* p1 = tile + 256 * (r + 4) + ((ch ^ swz256(r + 4)) << 4) + sub;
    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((fw_lds_s16x4*)p0));
    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((fw_lds_s16x4*)p1));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// XT = true : X token-major as well (dW = dY^T x; bias gradient by ones-MFMA).
// XT = false: X k-contiguous [M][K] (dX = dY W with W stored [K][N]): its tile is the 128-byte-row image of gemm_kernel
//             (global_load_lds + frag_sw), only W takes the transposing path.
template <bool XT>
__global__ __launch_bounds__(256) void gemm_tr_kernel(GemmArgs a) {
    using T = bf16raw;
    constexpr int KT = 64, WM = 4, TILE = 64 * 256;      // k per step, m tiles per wave, bytes per operand tile (= 128 * LDS_ROW)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto xs = [&](int i) -> char* { return smem + i * 2 * TILE; };
    auto ws = [&](int i) -> char* { return smem + i * 2 * TILE + TILE; };
    const int wave = threadIdx.x >> 6;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {                                                     // contiguous eighths of the (slice, n tile, m tile) order per XCD, see gemm_kernel
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (gridDim.z > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned w = xcd_contiguous(lin, total);
            bx = (int)(w % gridDim.x);
            by = (int)((w / gridDim.x) % gridDim.y);
            bz = (int)(w / (gridDim.x * gridDim.y));
        } else if (gridDim.z == 1 && gridDim.y > 1) {
            // no split-K: every XCD takes a contiguous eighth of a GROUPED tile order (bands of 8 m tiles, n tiles swept inside a
            // band), so the ~64 tiles an XCD has in flight touch 8 x 8 operand panels that fit its 4 MB L2 instead of re-fetching
            // X once per n tile (PMC: 2.5x the algorithmic bytes crossed the fabric in plain row-major order)
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned w = xcd_contiguous(lin, total);
            const unsigned per_band = 8 * gridDim.y;
            const unsigned band = w / per_band, first = band * 8;
            const unsigned gsz = min(gridDim.x - first, 8u);
            bx = (int)(first + (w % per_band) % gsz);
            by = (int)((w % per_band) / gsz);
        }
    }
    const int m_blk = bx * 128, n_blk = by * 128;
    const int wm0 = (wave & 1) * 64, wn0 = (wave >> 1) * 64;
    const int k_begin = bz * a.kper;
    const int k_end = min(a.K, k_begin + a.kper);
    const int nsteps = (k_end - k_begin) / KT;           // whole steps only (host guarantees K % KT == 0)

    f32x4 acc[4][WM], xsacc[WM];
    zero_acc(acc);
#pragma unroll
    for (int m = 0; m < WM; ++m) xsacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_xsum = XT && a.xsum != nullptr && by == 0 && wn0 == 0;    // wave-uniform
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);   // bf16 1.0 x 8

    auto issue = [&](int k0, int buf) {
        if constexpr (XT) glds_issue_km(a.X, a.ldx, m_blk, a.M, k0, xs(buf));
        else glds_issue<T, BM>(a.X, a.ldx, m_blk, a.M, k0 * 2, xs(buf));
        glds_issue_km(a.W, a.ldw, n_blk, a.N, k0, ws(buf));
    };
    if (nsteps > 0) issue(k_begin, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) issue(k_begin + (s + 1) * KT, cur ^ 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 af[4], bfr[WM];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = frag_tr256(ws(cur), wn0 + 16 * m, c);
#pragma unroll
            for (int n = 0; n < WM; ++n) bfr[n] = XT ? frag_tr256(xs(cur), wm0 + 16 * n, c) : frag_sw(xs(cur), wm0 + 16 * n, c);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
            if (do_xsum) {
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(xsacc[n], ones, bfr[n]);
            }
        }
        __syncthreads();                                 // also drains the in-flight global_load_lds (vmcnt(0))
    }
    const int l = lane_id();
    if (do_xsum && (l >> 4) == 0) {                      // every row of the ones-product holds the column sums: take row 0
#pragma unroll
        for (int n = 0; n < WM; ++n) {
            const int m = m_blk + wm0 + n * 16 + l;
            if (m < a.M) {
                if (a.xsum_zstride > 0) a.xsum[(long)bz * a.xsum_zstride + m] = xsacc[n][0];
                else atomicAdd(a.xsum + m, xsacc[n][0]);
            }
        }
    }
    f32x4 bias4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
        bias4[nt] = epi_bias(a, n0 < a.N ? n0 : 0, bz);
    }
#pragma unroll
    for (int mt = 0; mt < WM; ++mt) {
        const int m = m_blk + wm0 + mt * 16 + (l & 15);
        const int mc = m < a.M ? m : a.M - 1;
        const float rs = a.rowscale ? a.rowscale[mc / a.rows_per_scale] : 1.0f;
        uint4 ext[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
            epi_fetch<T>(a, ext[nt], mc, n0 < a.N ? n0 : 0, bz);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
            if (m < a.M && n0 < a.N) epi_apply<T>(a, bias4[nt], ext[nt], acc[nt][mt], m, n0, rs, bz);
        }
    }
}

// Epilogue of a 128 x (64 * 4 / WM ... ) wave tile: acc[nt][mt] holds C[m = .. + mt*16 + (l & 15)][n0 = .. + nt*16 + 4*(l >> 4) .. +3].
// PLAIN = true is the specialisation for the launches that need nothing but `+ bias` and a store (QKV / projection forward,
// input gradients, split-K partial tiles: most launches of a step): about 10 instructions per quad instead of the ~50 of the
// general form, whose run-time switches (activation, DropPath scale, residual, GELU twin, accumulate) cost as much VALU issue as
// the MFMAs of a 7-step K loop.
template <typename T, int WM, bool PLAIN>
FW_DEV void tile_epilogue(const GemmArgs& a, const f32x4 (&acc)[4][WM], int m_blk, int n_blk, int wm0, int wn0, int bz) {
    const int l = lane_id();
    f32x4 bias4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
        bias4[nt] = epi_bias(a, n0 < a.N ? n0 : 0, bz);
    }
    if constexpr (PLAIN) {
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            const int m = m_blk + wm0 + mt * 16 + (l & 15);
            if (m >= a.M) continue;
            if (a.out_f32) {
                float* cp = reinterpret_cast<float*>(a.C) + (long)bz * a.c_zstride + (long)m * a.ldc;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    if (n0 < a.N) *reinterpret_cast<f32x4*>(cp + n0) = acc[nt][mt] + bias4[nt];
                }
            } else {
                T* cp = reinterpret_cast<T*>(a.C) + (long)m * a.ldc;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    if (n0 < a.N) {
                        const f32x4 v = acc[nt][mt] + bias4[nt];
                        if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(cp + n0) = v;
                        else *reinterpret_cast<uint2*>(cp + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
                }
            }
        }
    } else {
        // Lean forms of the three epilogues that carry most of a step's non-plain launches: straight-line code, decided ONCE per
        // wave (the conditions are kernel arguments) instead of ~10 run-time switches per quad in the general form below.
        const bool tw = sizeof(T) == 2 && a.alpha == 1.0f && !a.out_f32 && !a.rowscale && !a.residual && !a.accumulate;
        if (tw && a.act == 2 && !a.C2) {                       // (A) dX = (dY W) o GELU'(aux): fc2's input gradient
            auto fetchA = [&](uint2 (&ext)[4], int mt) {
                const int m = m_blk + wm0 + mt * 16 + (l & 15);
                const T* ap = reinterpret_cast<const T*>(a.aux) + (long)(m < a.M ? m : a.M - 1) * a.ldaux;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    ext[nt] = *reinterpret_cast<const uint2*>(ap + (n0 < a.N ? n0 : 0));
                }
            };
            auto applyA = [&](const uint2 (&ext)[4], int mt) {
                const int m = m_blk + wm0 + mt * 16 + (l & 15);
                if (m >= a.M) return;
                T* cp = reinterpret_cast<T*>(a.C) + (long)m * a.ldc;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    if (n0 >= a.N) continue;
                    const f32x4 v = acc[nt][mt] + bias4[nt];
                    const float x0 = __uint_as_float(ext[nt].x << 16), x1 = __uint_as_float(ext[nt].x & 0xffff0000u);
                    const float x2 = __uint_as_float(ext[nt].y << 16), x3 = __uint_as_float(ext[nt].y & 0xffff0000u);
                    *reinterpret_cast<uint2*>(cp + n0) = make_uint2(pack_bf2(v[0] * gelu_grad_poly(x0), v[1] * gelu_grad_poly(x1)),
                                                                    pack_bf2(v[2] * gelu_grad_poly(x2), v[3] * gelu_grad_poly(x3)));
                }
            };
            uint2 ea[4], eb[4];
            fetchA(ea, 0);
#pragma unroll
            for (int mt = 0; mt < WM; mt += 2) {
                if (mt + 1 < WM) fetchA(eb, mt + 1);
                applyA(ea, mt);
                if (mt + 1 < WM) {
                    if (mt + 2 < WM) fetchA(ea, mt + 2);
                    applyA(eb, mt + 1);
                }
            }
            return;
        }
        if (tw && a.act == 0 && a.C2) {                        // (B) h = x W^T + b and its GELU twin: fc1 forward
#pragma unroll
            for (int mt = 0; mt < WM; ++mt) {
                const int m = m_blk + wm0 + mt * 16 + (l & 15);
                if (m >= a.M) continue;
                T* cp = reinterpret_cast<T*>(a.C) + (long)m * a.ldc;
                T* c2 = reinterpret_cast<T*>(a.C2) + (long)m * a.ldc2;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    if (n0 >= a.N) continue;
                    const f32x4 v = acc[nt][mt] + bias4[nt];
                    *reinterpret_cast<uint2*>(cp + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    *reinterpret_cast<uint2*>(c2 + n0) = make_uint2(pack_bf2(gelu_poly(v[0]), gelu_poly(v[1])), pack_bf2(gelu_poly(v[2]), gelu_poly(v[3])));
                }
            }
            return;
        }
        if (a.act == 0 && a.residual && a.out_f32 && !a.C2 && a.alpha == 1.0f && !a.accumulate && a.splitk == 1 && a.c_zstride == 0) {
            // (C) y = res + rs * (x W^T + b), f32 out: the attention projection and fc2 forward (DropPath row scale optional)
            auto fetchC = [&](f32x4 (&ext)[4], int mt) {
                const int m = m_blk + wm0 + mt * 16 + (l & 15);
                const float* rp = a.residual + (long)(m < a.M ? m : a.M - 1) * a.ldr;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    ext[nt] = *reinterpret_cast<const f32x4*>(rp + (n0 < a.N ? n0 : 0));
                }
            };
            auto applyC = [&](const f32x4 (&ext)[4], int mt) {
                const int m = m_blk + wm0 + mt * 16 + (l & 15);
                if (m >= a.M) return;
                const float rs = a.rowscale ? a.rowscale[m / a.rows_per_scale] : 1.0f;
                float* cp = reinterpret_cast<float*>(a.C) + (long)m * a.ldc;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                    if (n0 < a.N) *reinterpret_cast<f32x4*>(cp + n0) = (acc[nt][mt] + bias4[nt]) * rs + ext[nt];
                }
            };
            f32x4 ea[4], eb[4];
            fetchC(ea, 0);
#pragma unroll
            for (int mt = 0; mt < WM; mt += 2) {
                if (mt + 1 < WM) fetchC(eb, mt + 1);
                applyC(ea, mt);
                if (mt + 1 < WM) {
                    if (mt + 2 < WM) fetchC(ea, mt + 2);
                    applyC(eb, mt + 1);
                }
            }
            return;
        }
        // the row-dependent operand (GELU' input / residual) of 16-row group mt + 1 is requested before group mt is applied and
        // stored: one exposed memory latency per tile instead of one per group.  (Requesting the whole wave tile up front -- 16
        // loads, 64 registers -- measured SLOWER than group-by-group: 135 -> 145 us at 16384 x 1792 x 448.)
        auto fetch = [&](uint4 (&ext)[4], int mt) {
            const int m = m_blk + wm0 + mt * 16 + (l & 15);
            const int mc = m < a.M ? m : a.M - 1;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                epi_fetch<T>(a, ext[nt], mc, n0 < a.N ? n0 : 0, bz);
            }
        };
        auto apply = [&](const uint4 (&ext)[4], int mt) {
            const int m = m_blk + wm0 + mt * 16 + (l & 15);
            const int mc = m < a.M ? m : a.M - 1;
            const float rs = a.rowscale ? a.rowscale[mc / a.rows_per_scale] : 1.0f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
                if (m < a.M && n0 < a.N) epi_apply<T>(a, bias4[nt], ext[nt], acc[nt][mt], m, n0, rs, bz);
            }
        };
        uint4 ea[4], eb[4];
        fetch(ea, 0);
#pragma unroll
        for (int mt = 0; mt < WM; mt += 2) {
            if (mt + 1 < WM) fetch(eb, mt + 1);
            apply(ea, mt);
            if (mt + 1 < WM) {
                if (mt + 2 < WM) fetch(ea, mt + 2);
                apply(eb, mt + 1);
            }
        }
    }
}
static inline bool plain_epilogue(const GemmArgs& a) {
    return a.act == 0 && !a.rowscale && !a.residual && !a.C2 && a.alpha == 1.0f && !(a.accumulate && a.c_zstride == 0);
}

// ---- the same product with a DEEP global -> LDS pipeline ------------------------------------------------------------
// gemm_tr_kernel above keeps ONE stage in flight per workgroup and drains it (`__syncthreads()` = vmcnt(0)) every K step: with
// two workgroups per CU a step costs one L2 / fabric round trip (1.4 us for 32 KB measured in the B = 16 step), i.e. the kernel
// is bound by LATENCY, not by bytes or MFMAs.  Here the stages form a ring of NS LDS buffers: NS-1 stages are in flight, a wave
// waits only for ITS loads of the oldest one (counted `s_waitcnt vmcnt(N)`: loads complete in issue order), a raw `s_barrier`
// then makes every wave's share of that stage visible, the buffer freed by the previous step is re-issued at once, and the
// MFMAs of the stage run while the younger stages keep arriving.  One barrier per stage; WAR is covered by the same barrier
// (a wave reaches it only after the MFMAs that consumed its fragments of the previous stage).
// hipcc (ROCm 7.2) cannot tell which LDS bytes an in-flight global_load_lds will write, so it puts `s_waitcnt vmcnt(0)` in front
// of the first ds_read that follows one (it does so in gemm_tr_kernel / gemm_kernel above: their "prefetch" of the next stage is
// waited for BEFORE the current stage is computed -- no overlap inside a workgroup).  The ring therefore issues its LDS-DMA from
// inline asm, which the compiler neither counts nor waits for, and counts completion itself (cdna_hip_programming.md 5.7):
// M0 carries the wave-uniform LDS destination and is written in the same statement that uses it.
template <int N> FW_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
FW_DEV void glds16_asm(const char* gsrc, char* lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// Source pointers of a thread's LDS-DMA instructions for stage 0, computed ONCE (row * ld + swizzled column: 64-bit multiplies);
// a later stage only adds its K offset.  (Recomputed per stage, the address arithmetic was ~15 VALU instructions per 1 KB moved.)
// Token-major operand: one [KT tokens][128 cols] tile, a wave-instruction fills 4 token rows (1 KB, lane-linear).
template <int KT> struct GldsKm {
    static constexpr int NI = KT / 16;
    const char* src[NI];
    long kstride;                                        // bytes per token row
    int loff[NI];                                        // LDS byte offset of the instruction's 1 KB inside the tile
    FW_MEM void init(const char* base, long ld, int col0, int cols_total, int k0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        kstride = ld * 2;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int R0 = wave * (KT / 4) + it * 4;
            const int r = R0 + (lane >> 4), p = lane & 15;
            int col = col0 + ((p ^ swz256(r)) << 3);
            if (col >= cols_total) col = 0;                // columns past the matrix only feed masked outputs; stay inside the row
            src[it] = base + ((long)(k0 + r) * ld + col) * 2;
            loff[it] = R0 * 256;
        }
    }
    FW_MEM void issue(int step, char* tile) const {
        const long off = (long)step * KT * kstride;
#pragma unroll
        for (int it = 0; it < NI; ++it) glds16_asm(src[it] + off, tile + loff[it]);
    }
};
// k-contiguous operand: the [ROWS][128 B] image of gemm_kernel (slot swizzle on the source address)
template <typename T, int ROWS> struct GldsKc {
    static constexpr int NI = ROWS / 32;
    const char* src[NI];
    int loff[NI];
    FW_MEM void init(const char* base, long ld, int row0, int rows_total, int kbyte0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int R0 = (wave * NI + it) * 8;
            const int r = R0 + (lane >> 3), p = lane & 7;
            int gr = row0 + r;
            if (gr >= rows_total) gr = rows_total - 1;     // rows past the end only feed masked outputs
            src[it] = base + (long)gr * ld * TT<T>::SZ + kbyte0 + ((p ^ (swz(r) >> 4)) << 4);
            loff[it] = R0 * LDS_ROW;
        }
    }
    FW_MEM void issue(int step, char* tile) const {
#pragma unroll
        for (int it = 0; it < NI; ++it) glds16_asm(src[it] + (long)step * 128, tile + loff[it]);
    }
};

constexpr int ROW64 = 64;
FW_DEV int swz64(int r) { return (r >> 2) & 3; }
template <typename T, int ROWS> struct GldsKc64 {
    static constexpr int NI = ROWS / 64;                     // wave instructions per thread and stage (4 waves x 16 rows each)
    const char* src[NI];
    int loff[NI];
    FW_MEM void init(const char* base, long ld, int row0, int rows_total, int kbyte0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int R0 = (wave * NI + it) * 16;
            const int r = R0 + (lane >> 2), p = lane & 3;
            int gr = row0 + r;
            if (gr >= rows_total) gr = rows_total - 1;     // rows past the end only feed masked outputs
            src[it] = base + (long)gr * ld * TT<T>::SZ + kbyte0 + ((p ^ swz64(r)) << 4);
            loff[it] = R0 * ROW64;
        }
    }
    FW_MEM void issue(int step, char* tile) const {
#pragma unroll
        for (int it = 0; it < NI; ++it) glds16_asm(src[it] + (long)step * ROW64, tile + loff[it]);
    }
};
FW_DEV uint4 frag_sw64(const char* tile, int row0) {
    const int l = lane_id();
    const int row = row0 + (l & 15);
    return *reinterpret_cast<const uint4*>(tile + row * ROW64 + (((l >> 4) ^ swz64(row)) << 4));
}

// ---- bf16 epilogue through LDS: full 128-byte row segments instead of 8-byte pieces ---------------------------------------------------
// The MFMA C/D layout hands a lane 4 consecutive n of ONE row: stored directly that is an 8-byte store per lane, 16 rows x 32 bytes per
// wave instruction -- the store path takes them at ~7 B/clk/CU (cdna_hip_programming.md T21), and with the operand loads removed the
// 256 x 256 kernel still spent 39 of its 71 us in this epilogue (16384 x 1792 x 448 with the GELU twin: tools/big_gemm_bench.py,
// FW_GEMM_BIG_DBG).  Here a wave parks its [16 * WM rows][64 columns] sub-tile in LDS (its own region: no workgroup barrier; rows
// padded to 136 bytes, so the 16 lanes of a column group fall on 32 distinct banks) and writes it out as whole 128-byte row segments,
// 16 bytes per lane, 8 rows per wave instruction.  mode 0: v = acc + bias;  1: also the GELU twin into C2;  2: v * GELU'(aux), the
// aux operand fetched the same way (row segments -> LDS -> the lane's quad).  Needs bf16 C, N % 8 == 0, ldc % 8 == 0.
constexpr int EPI_LD = 136;
template <int WM>
FW_DEV void tile_epilogue_staged(const GemmArgs& a, const f32x4 (&acc)[4][WM], int m_blk, int n_blk, int wm0, int wn0, char* stage, int mode) {
    const int l = lane_id();
    char* mine = stage + (threadIdx.x >> 6) * (16 * WM * EPI_LD);
    f32x4 bias4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n0 = n_blk + wn0 + nt * 16 + ((l >> 4) << 2);
        bias4[nt] = epi_bias(a, n0 < a.N ? n0 : 0, 0);
    }
    const int rsub = l >> 3, cb = (l & 7) * 16;                   // row-segment form: lane = (row in a group of 8, 16-byte chunk of the 128-byte segment)
    const int ncol = n_blk + wn0 + (l & 7) * 8;                   // first of the lane's 8 columns
    const bool col_ok = ncol < a.N;
    auto rows_out = [&](char* C, long ldc) {                      // LDS region -> global, whole row segments
#pragma unroll
        for (int it = 0; it < 2 * WM; ++it) {
            const int r = it * 8 + rsub, m = m_blk + wm0 + r;
            const uint4 v = *reinterpret_cast<const uint4*>(mine + r * EPI_LD + cb);
            if (m < a.M && col_ok) *reinterpret_cast<uint4*>(C + ((long)m * ldc + ncol) * 2) = v;
        }
    };
    if (mode == 2) {                                              // aux rows in: global row segments -> LDS
#pragma unroll
        for (int it = 0; it < 2 * WM; ++it) {
            const int r = it * 8 + rsub, m = m_blk + wm0 + r;
            const long mc = m < a.M ? m : a.M - 1;
            const uint4 v = *reinterpret_cast<const uint4*>(a.aux + (mc * a.ldaux + (col_ok ? ncol : 0)) * 2);
            *reinterpret_cast<uint4*>(mine + r * EPI_LD + cb) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && mode != 1) break;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) {
            char* rowp = mine + (mt * 16 + (l & 15)) * EPI_LD + ((l >> 4) << 3);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 v = acc[nt][mt] + bias4[nt];
                if (mode == 2) {
                    const uint2 e = *reinterpret_cast<const uint2*>(rowp + nt * 32);
                    v[0] *= gelu_grad_poly(__uint_as_float(e.x << 16)); v[1] *= gelu_grad_poly(__uint_as_float(e.x & 0xffff0000u));
                    v[2] *= gelu_grad_poly(__uint_as_float(e.y << 16)); v[3] *= gelu_grad_poly(__uint_as_float(e.y & 0xffff0000u));
                } else if (pass == 1) {
                    v[0] = gelu_poly(v[0]); v[1] = gelu_poly(v[1]); v[2] = gelu_poly(v[2]); v[3] = gelu_poly(v[3]);
                }
                *reinterpret_cast<uint2*>(rowp + nt * 32) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (pass == 0) rows_out(a.C, a.ldc); else rows_out(a.C2, a.ldc2);
        if (pass == 0 && mode == 1) {                             // the twin pass rewrites the region: the reads above must have returned
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
// which staged mode serves this launch (-1: none)
static inline int staged_mode(const GemmArgs& a) {
    if (a.out_f32 || a.alpha != 1.0f || a.rowscale || a.residual || a.accumulate || a.c_zstride || a.splitk != 1 || a.N % 8 || a.ldc % 8) return -1;
    if (((uintptr_t)a.C & 15)) return -1;
    if (a.act == 0 && !a.C2) return 0;
    if (a.act == 0 && a.C2) return (a.ldc2 % 8 || ((uintptr_t)a.C2 & 15)) ? -1 : 1;
    if (a.act == 2 && !a.C2) return (a.ldaux % 8 || ((uintptr_t)a.aux & 15)) ? -1 : 2;
    return -1;
}

// one 128 x 128 output tile (bx, by) of K slice bz; `smem`: NS stages.  Shared by the per-problem kernel and the grouped one.
template <bool XT, int KT, int NS, bool PLAIN>
FW_DEV void tr_ring_tile(const GemmArgs& a, int bx, int by, int bz, char* smem) {
    using T = bf16raw;
    static_assert(NS >= 2 && NS <= 5, "ring depth");
    constexpr int WM = 4;
    constexpr bool X64 = !XT && KT == 32;                      // k-contiguous X in 64-byte rows (GldsKc64): 32-deep steps, e.g. K = 224
    constexpr int XB = XT ? KT * 256 : 128 * (X64 ? ROW64 : LDS_ROW), WB = KT * 256, STAGE = XB + WB;
    constexpr int LPS = (XT ? KT / 16 : (X64 ? 2 : 4)) + KT / 16;          // global_load_lds instructions per thread and stage
    auto xs = [&](int i) -> char* { return smem + i * STAGE; };
    auto ws = [&](int i) -> char* { return smem + i * STAGE + XB; };
    const int wave = threadIdx.x >> 6;
    const int m_blk = bx * 128, n_blk = by * 128;
    const int wm0 = (wave & 1) * 64, wn0 = (wave >> 1) * 64;
    const int k_begin = bz * a.kper;
    const int k_end = min(a.K, k_begin + a.kper);
    const int nsteps = (k_end - k_begin) / KT;           // whole steps only (host guarantees K % KT == 0)

    f32x4 acc[4][WM], xsacc[WM];
    zero_acc(acc);
#pragma unroll
    for (int m = 0; m < WM; ++m) xsacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_xsum = XT && a.xsum != nullptr && by == 0 && wn0 == 0;    // wave-uniform
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);   // bf16 1.0 x 8

    GldsKm<KT> gw;
    gw.init(a.W, a.ldw, n_blk, a.N, k_begin);
    GldsKm<XT ? KT : 16> gxm;
    GldsKc<T, BM> gxc;
    GldsKc64<T, BM> gxc64;
    if constexpr (XT) gxm.init(a.X, a.ldx, m_blk, a.M, k_begin);
    else if constexpr (X64) gxc64.init(a.X, a.ldx, m_blk, a.M, k_begin * 2);
    else gxc.init(a.X, a.ldx, m_blk, a.M, k_begin * 2);
    auto issue = [&](int step, int buf) {
        if constexpr (XT) gxm.issue(step, xs(buf));
        else if constexpr (X64) gxc64.issue(step, xs(buf));
        else gxc.issue(step, xs(buf));
        gw.issue(step, ws(buf));
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nsteps) issue(p, p);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        // stages s+1 .. s+NS-2 may stay in flight; near the tail fewer were issued
        const int younger = min(NS - 2, nsteps - 1 - s);
        if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
        else if (NS >= 4 && younger == NS - 3) wait_vmcnt<(NS >= 4 ? NS - 3 : 0) * LPS>();
        else if (NS >= 5 && younger == NS - 4) wait_vmcnt<(NS >= 5 ? NS - 4 : 0) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nsteps) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);   // the buffer step s-1 has just released
#pragma unroll
        for (int c = 0; c < KT / 32; ++c) {
            uint4 af[4], bfr[WM];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = frag_tr256(ws(buf), wn0 + 16 * m, c);
#pragma unroll
            for (int n = 0; n < WM; ++n) bfr[n] = XT ? frag_tr256(xs(buf), wm0 + 16 * n, c) : (X64 ? frag_sw64(xs(buf), wm0 + 16 * n) : frag_sw(xs(buf), wm0 + 16 * n, c));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
            if (do_xsum) {
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(xsacc[n], ones, bfr[n]);
            }
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    const int l = lane_id();
    if (do_xsum && (l >> 4) == 0) {                      // every row of the ones-product holds the column sums: take row 0
#pragma unroll
        for (int n = 0; n < WM; ++n) {
            const int m = m_blk + wm0 + n * 16 + l;
            if (m < a.M) {
                if (a.xsum_zstride > 0) a.xsum[(long)bz * a.xsum_zstride + m] = xsacc[n][0];
                else atomicAdd(a.xsum + m, xsacc[n][0]);
            }
        }
    }
    if (a.staged >= 0 && (size_t)NS * STAGE >= (size_t)4 * 16 * WM * EPI_LD) {
        __syncthreads();                                   // every wave has left the ring: its space stages the output rows
        tile_epilogue_staged<WM>(a, acc, m_blk, n_blk, wm0, wn0, smem, a.staged);
        return;
    }
    tile_epilogue<T, WM, PLAIN>(a, acc, m_blk, n_blk, wm0, wn0, bz);
}

template <bool XT, int KT, int NS, bool PLAIN>
__global__ __launch_bounds__(256, (KT == 32 && NS == 3) ? 3 : 2) void gemm_tr_ring_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {                                                     // contiguous eighths of the (slice, n tile, m tile) order per XCD, see gemm_kernel
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (gridDim.z > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned w = xcd_contiguous(lin, total);
            bx = (int)(w % gridDim.x);
            by = (int)((w / gridDim.x) % gridDim.y);
            bz = (int)(w / (gridDim.x * gridDim.y));
        } else if (gridDim.z == 1 && gridDim.y > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned w = xcd_contiguous(lin, total);
            const unsigned per_band = 8 * gridDim.y;
            const unsigned band = w / per_band, first = band * 8;
            const unsigned gsz = min(gridDim.x - first, 8u);
            bx = (int)(first + (w % per_band) % gsz);
            by = (int)((w % per_band) / gsz);
        }
    }
    tr_ring_tile<XT, KT, NS, PLAIN>(a, bx, by, bz, smem);
}

// ---- every weight gradient of a backward pass in ONE launch ------------------------------------------------------------------------
// A backward pass of the B = 16 step holds 251 products dW = dY^T x (82 distinct shapes).  Launched one by one, each has to fill the
// chip on its own: the deep stages (C >= 224: 50 .. 200 output tiles, 4 096 .. 16 384 tokens) were split 8 .. 32 ways over tokens into
// f32 partial slabs -- 5.9 GB of slab write + re-read per step (profiles/r02: slab_reduce_multi 6.15 GB, 1.14 ms) -- and every launch
// paid its own ramp and tail.  Nobody reads a weight gradient before the optimizer step, so the host queues them (fwair/ops.py:
// wgrad(defer=True)) and the end-of-pass callback launches this kernel ONCE over a table of tiles: thousands of work items keep every
// CU busy without splitting the deep stages at all (a tile owns its whole token range and adds straight into the gradient), only the
// tall reductions (tokens > FW_WGRAD_CHUNK) are cut into slices whose small partial tiles go through the slab fold as before.
// probs: GemmArgs per product (device memory);  items: {problem, m tile, n tile, slice} in execution order (the host lays the
// logical order out so that each XCD runs a contiguous eighth: tiles sharing operand rows meet in one L2).
__global__ __launch_bounds__(256, 3) void gemm_wgrad_group_kernel(const GemmArgs* __restrict__ probs, const int4* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int4 it = items[blockIdx.x];
    if (it.x < 0) return;                                  // padding of an XCD's work list
    const GemmArgs a = probs[it.x];
    if (a.c_zstride > 0) tr_ring_tile<true, 32, 3, true>(a, it.y, it.z, it.w, smem);      // a slice's partial tile: plain store into its slab
    else tr_ring_tile<true, 32, 3, false>(a, it.y, it.z, it.w, smem);                     // the whole reduction: dW += tile (f32 read-modify-write)
}

// The same product on 256 x 256 tiles (8 waves, wave tile 128 x 64, 32-token steps, 3 stages of 32 KB): half the operand bytes per
// FLOP of the 128 x 128 form.  Both operands token-major: two [32][128] images each, read with ds_read_b64_tr_b16.
// Waves 0-3 fill the X images (2 waves x 16 token rows per image), waves 4-7 the W images: 4 LDS-DMA instructions per thread and stage.
template <bool PLAIN>
FW_DEV void tr_big_tile(const GemmArgs& a, int bx, int by, int bz, char* smem) {
    using T = bf16raw;
    constexpr int KT = 32, NS = 3, WM = 8, IMG = KT * 256, STAGE = 4 * IMG, LPS = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m_blk = bx * 256, n_blk = by * 256;
    const int wm0 = (wave & 1) * 128, wn0 = (wave >> 1) * 64;
    const int k_begin = bz * a.kper;
    const int k_end = min(a.K, k_begin + a.kper);
    const int nsteps = (k_end - k_begin) / KT;
    const char* src[4]; int off[4];
    {
        const bool isw = wave >= 4;
        const int half = (wave >> 1) & 1, rg = wave & 1;
        const char* base = isw ? a.W : a.X;
        const long ld = isw ? a.ldw : a.ldx;
        const int col0 = (isw ? n_blk : m_blk) + half * 128, cols = isw ? a.N : a.M;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int R0 = rg * 16 + it * 4, r = R0 + (lane >> 4), p = lane & 15;
            int col = col0 + ((p ^ swz256(r)) << 3);
            if (col >= cols) col = 0;
            src[it] = base + ((long)(k_begin + r) * ld + col) * 2;
            off[it] = (isw ? 2 * IMG : 0) + half * IMG + R0 * 256;
        }
    }
    const long kstride = (wave >= 4 ? a.ldw : a.ldx) * 2;
    auto issue = [&](int step, int buf) {
        char* st = smem + buf * STAGE;
#pragma unroll
        for (int it = 0; it < 4; ++it) glds16_asm(src[it] + (long)step * KT * kstride, st + off[it]);
    };
    f32x4 acc[4][WM];
    zero_acc(acc);
    // bias gradient (column sums of X = dY): the lane's 8 tokens of a fragment summed with v_dot2c_f32_bf16 against ones -- 8 registers
    // instead of the 32 a ones-MFMA accumulator per column group costs (that form spilled: 128 accumulators + 48 fragment registers)
    float xs[WM];
#pragma unroll
    for (int m = 0; m < WM; ++m) xs[m] = 0.f;
    const bool do_xsum = a.xsum != nullptr && by == 0 && wn0 == 0;        // wave-uniform
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    const bf2_t ones2 = __builtin_bit_cast(bf2_t, 0x3F803F80u);
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nsteps) issue(p, p);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int younger = min(NS - 2, nsteps - 1 - s);
        if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nsteps) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const char* st = smem + buf * STAGE;
        uint4 af[4], bfr[WM];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = frag_tr256(st + 2 * IMG + (wn0 >> 7) * IMG, (wn0 & 127) + 16 * m, 0);
#pragma unroll
        for (int n = 0; n < WM; ++n) bfr[n] = frag_tr256(st + (wm0 >> 7) * IMG, 16 * n, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
        if (do_xsum) {
#pragma unroll
            for (int n = 0; n < WM; ++n) {
                xs[n] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, bfr[n].x), ones2, xs[n], false);
                xs[n] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, bfr[n].y), ones2, xs[n], false);
                xs[n] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, bfr[n].z), ones2, xs[n], false);
                xs[n] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, bfr[n].w), ones2, xs[n], false);
            }
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    const int l = lane_id();
    if (do_xsum) {                                       // lane (l & 15) = column, the four lane groups hold disjoint token subsets
#pragma unroll
        for (int n = 0; n < WM; ++n) {
            float v = xs[n];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            const int m = m_blk + wm0 + n * 16 + l;
            if ((l >> 4) == 0 && m < a.M) {
                if (a.xsum_zstride > 0) a.xsum[(long)bz * a.xsum_zstride + m] = v;
                else atomicAdd(a.xsum + m, v);
            }
        }
    }
    tile_epilogue<T, WM, PLAIN>(a, acc, m_blk, n_blk, wm0, wn0, bz);
}
__global__ __launch_bounds__(512) void gemm_wgrad_group_big_kernel(const GemmArgs* __restrict__ probs, const int4* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int4 it = items[blockIdx.x];
    if (it.x < 0) return;
    const GemmArgs a = probs[it.x];
    if (a.c_zstride > 0) tr_big_tile<true>(a, it.y, it.z, it.w, smem);
    else tr_big_tile<false>(a, it.y, it.z, it.w, smem);
}

template <bool XT, int KT, int NS, bool PLAIN>
int launch_tr_ring_p(const GemmArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NS * ((XT ? KT * 256 : 128 * ((KT == 32) ? ROW64 : LDS_ROW)) + KT * 256);
    FW_SET_LDS_ONCE((gemm_tr_ring_kernel<XT, KT, NS, PLAIN>), lds);
    FW_KNAME("gemm_tr_ring_kernel<%s,%d,%d,%s>", FW_B(XT), KT, NS, FW_B(PLAIN));
    hipLaunchKernelGGL((gemm_tr_ring_kernel<XT, KT, NS, PLAIN>), dim3(fw_cdiv(a.M, 128), fw_cdiv(a.N, 128), a.splitk), dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}
template <bool XT, int KT, int NS>
int launch_tr_ring(const GemmArgs& a, hipStream_t st) {
    return plain_epilogue(a) ? launch_tr_ring_p<XT, KT, NS, true>(a, st) : launch_tr_ring_p<XT, KT, NS, false>(a, st);
}

// ---- gemm_kernel's NT product (both operands k-contiguous, whole 128-byte K steps) on the same ring ---------------------
// Forward Linears of the C >= 224 stages, im2col / pixel-shuffle convolutions: y = x W^T.  Stage = [128 + BN rows][128 B].
// BN = 128: 4 stages of 32 KB (1 workgroup per CU, 96 KB in flight); BN = 64: 3 stages of 24 KB (2 workgroups per CU).
template <typename T, int BN, int NS, bool PLAIN>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmArgs a) {
    constexpr int KT = 128 / TT<T>::SZ;
    constexpr int WM = (BN == 128) ? 4 : 2;
    constexpr int XBYTES = BM * LDS_ROW, WBYTES = BN * LDS_ROW, STAGE = XBYTES + WBYTES;
    constexpr int LPS = BM / 32 + BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto xs = [&](int i) -> char* { return smem + i * STAGE; };
    auto ws = [&](int i) -> char* { return smem + i * STAGE + XBYTES; };
    const int wave = threadIdx.x >> 6;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (gridDim.z > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned w = xcd_contiguous(lin, total);
            bx = (int)(w % gridDim.x);
            by = (int)((w / gridDim.x) % gridDim.y);
            bz = (int)(w / (gridDim.x * gridDim.y));
        } else if (gridDim.z == 1 && gridDim.y > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned w = xcd_contiguous(lin, total);
            const unsigned per_band = 8 * gridDim.y;
            const unsigned band = w / per_band, first = band * 8;
            const unsigned gsz = min(gridDim.x - first, 8u);
            bx = (int)(first + (w % per_band) % gsz);
            by = (int)((w % per_band) / gsz);
        }
    }
    const int m_blk = bx * BM, n_blk = by * BN;
    const int wm0 = (BN == 128) ? (wave & 1) * 64 : wave * 32;
    const int wn0 = (BN == 128) ? (wave >> 1) * 64 : 0;
    const int k_begin = bz * a.kper;
    const int k_end = min(a.K, k_begin + a.kper);
    const int nsteps = (k_end - k_begin) / KT;           // whole steps (host: K % KT == 0)

    f32x4 acc[4][WM];
    zero_acc(acc);
    GldsKc<T, BM> gx;
    GldsKc<T, BN> gw;
    gx.init(a.X, a.ldx, m_blk, a.M, k_begin * TT<T>::SZ);
    gw.init(a.W, a.ldw, n_blk, a.N, k_begin * TT<T>::SZ);
    auto issue = [&](int step, int buf) {
        gx.issue(step, xs(buf));
        gw.issue(step, ws(buf));
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nsteps) issue(p, p);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int younger = min(NS - 2, nsteps - 1 - s);
        if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
        else if (NS >= 4 && younger == NS - 3) wait_vmcnt<(NS >= 4 ? NS - 3 : 0) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nsteps) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint4 af[4], bfr[WM];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = frag_sw(ws(buf), wn0 + 16 * m, c);
#pragma unroll
            for (int n = 0; n < WM; ++n) bfr[n] = frag_sw(xs(buf), wm0 + 16 * n, c);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    if constexpr (sizeof(T) == 2) {
        if (a.staged >= 0 && (size_t)NS * STAGE >= (size_t)4 * 16 * WM * EPI_LD) {
            __syncthreads();
            tile_epilogue_staged<WM>(a, acc, m_blk, n_blk, wm0, wn0, smem, a.staged);
            return;
        }
    }
    tile_epilogue<T, WM, PLAIN>(a, acc, m_blk, n_blk, wm0, wn0, bz);
}

template <typename T, int BN, int NS, bool PLAIN>
int launch_ring_p(const GemmArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NS * (BM + BN) * LDS_ROW;
    FW_SET_LDS_ONCE((gemm_ring_kernel<T, BN, NS, PLAIN>), lds);
    dim3 grid(fw_cdiv(a.M, BM), fw_cdiv(a.N, BN), a.splitk);
    FW_KNAME("gemm_ring_kernel<%s,%d,%d,%s>", tname<T>(), BN, NS, FW_B(PLAIN));
    hipLaunchKernelGGL((gemm_ring_kernel<T, BN, NS, PLAIN>), grid, dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}
template <typename T, int BN, int NS>
int launch_ring(const GemmArgs& a, hipStream_t st) {
    return plain_epilogue(a) ? launch_ring_p<T, BN, NS, true>(a, st) : launch_ring_p<T, BN, NS, false>(a, st);
}

// ---- the NT product on 64-BYTE K steps: four 16 KB stages in the same 64 KB (BN = 128), three loads in flight ----------------
// gemm_ring_kernel above holds 2 x 32 KB stages: one stage in flight per workgroup, 62 % of its wave cycles parked in s_waitcnt
// (SQ_WAIT_ANY).  The weight-gradient kernel -- four 16 KB stages of 32-deep steps in the same LDS, two workgroups per CU -- waits
// 33 %.  This is that shape for k-contiguous operands: LDS rows of 64 bytes, 16-byte slot p of row r stored at slot
// p ^ ((r >> 2) & 3) (16 consecutive rows x one slot = 64 distinct banks for the fragment's ds_read_b128), filled by LDS-DMA with
// the swizzle applied to the per-lane SOURCE address (one wave instruction = 16 rows x 64 B = 1 KB, lane-linear in LDS).
// Also takes K that is a multiple of 64 bytes but not of 128 (K = 224, 672 in bf16).
template <typename T, int BN, int NS, bool PLAIN>
__global__ __launch_bounds__(256) void gemm_ring64_kernel(GemmArgs a) {
    constexpr int KT = ROW64 / TT<T>::SZ;
    constexpr int WM = (BN == 128) ? 4 : 2;
    constexpr int XBYTES = BM * ROW64, WBYTES = BN * ROW64, STAGE = XBYTES + WBYTES;
    constexpr int LPS = BM / 64 + BN / 64;
    static_assert(NS >= 3 && NS <= 5, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto xs = [&](int i) -> char* { return smem + i * STAGE; };
    auto ws = [&](int i) -> char* { return smem + i * STAGE + XBYTES; };
    const int wave = threadIdx.x >> 6;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (gridDim.z > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned w = xcd_contiguous(lin, total);
            bx = (int)(w % gridDim.x);
            by = (int)((w / gridDim.x) % gridDim.y);
            bz = (int)(w / (gridDim.x * gridDim.y));
        } else if (gridDim.z == 1 && gridDim.y > 1) {
            const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
            const unsigned w = xcd_contiguous(lin, total);
            const unsigned per_band = 8 * gridDim.y;
            const unsigned band = w / per_band, first = band * 8;
            const unsigned gsz = min(gridDim.x - first, 8u);
            bx = (int)(first + (w % per_band) % gsz);
            by = (int)((w % per_band) / gsz);
        }
    }
    const int m_blk = bx * BM, n_blk = by * BN;
    const int wm0 = (BN == 128) ? (wave & 1) * 64 : wave * 32;
    const int wn0 = (BN == 128) ? (wave >> 1) * 64 : 0;
    const int k_begin = bz * a.kper;
    const int k_end = min(a.K, k_begin + a.kper);
    const int nsteps = (k_end - k_begin) / KT;           // whole steps (host: K and kper are multiples of KT)

    f32x4 acc[4][WM];
    zero_acc(acc);
    GldsKc64<T, BM> gx;
    GldsKc64<T, BN> gw;
    gx.init(a.X, a.ldx, m_blk, a.M, k_begin * TT<T>::SZ);
    gw.init(a.W, a.ldw, n_blk, a.N, k_begin * TT<T>::SZ);
    auto issue = [&](int step, int buf) {
        gx.issue(step, xs(buf));
        gw.issue(step, ws(buf));
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nsteps) issue(p, p);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int younger = min(NS - 2, nsteps - 1 - s);
        if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
        else if (younger == NS - 3) wait_vmcnt<(NS - 3) * LPS>();
        else if (NS >= 5 && younger == NS - 4) wait_vmcnt<(NS >= 5 ? NS - 4 : 0) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nsteps) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        uint4 af[4], bfr[WM];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = frag_sw64(ws(buf), wn0 + 16 * m);
#pragma unroll
        for (int n = 0; n < WM; ++n) bfr[n] = frag_sw64(xs(buf), wm0 + 16 * n);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    if constexpr (sizeof(T) == 2) {
        if (a.staged >= 0 && (size_t)NS * STAGE >= (size_t)4 * 16 * WM * EPI_LD) {
            __syncthreads();
            tile_epilogue_staged<WM>(a, acc, m_blk, n_blk, wm0, wn0, smem, a.staged);
            return;
        }
    }
    tile_epilogue<T, WM, PLAIN>(a, acc, m_blk, n_blk, wm0, wn0, bz);
}

template <typename T, int BN, int NS, bool PLAIN>
int launch_ring64_p(const GemmArgs& a, hipStream_t st) {
    const size_t lds = (size_t)NS * (BM + BN) * ROW64;
    FW_SET_LDS_ONCE((gemm_ring64_kernel<T, BN, NS, PLAIN>), lds);
    dim3 grid(fw_cdiv(a.M, BM), fw_cdiv(a.N, BN), a.splitk);
    FW_KNAME("gemm_ring64_kernel<%s,%d,%d,%s>", tname<T>(), BN, NS, FW_B(PLAIN));
    hipLaunchKernelGGL((gemm_ring64_kernel<T, BN, NS, PLAIN>), grid, dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}
template <typename T, int BN, int NS>
int launch_ring64(const GemmArgs& a, hipStream_t st) {
    return plain_epilogue(a) ? launch_ring64_p<T, BN, NS, true>(a, st) : launch_ring64_p<T, BN, NS, false>(a, st);
}

// ---- 256 x 256 tiles, 8 waves: the compute-bound products (C = 448 / 896 stages, the 65 536-wide encoder heads) --------------------
// The 128 x 128 rings above move (128 + 128) x K operand bytes per 128 x 128 outputs through the CU's L2 port; at K = 448 .. 896 and
// N >= 1344 they sit at 350 .. 500 TFLOP/s with ~28 GB/s per CU of operand traffic -- bound by bytes in flight per CU, not by the
// MFMA pipe (1.5 us of MFMA per 21 us tile).  A 256 x 256 tile does four times the products on twice the bytes: wave w (of 8,
// two per SIMD) owns 128 rows (m) x 64 columns (n) = 32 accumulator tiles (128 VGPRs), a K chunk is 12 fragment reads for 32 MFMAs
// (16 for 32 before).  Same ring discipline as gemm_ring_kernel (LDS-DMA from inline asm, counted vmcnt, one raw barrier per stage):
//   KT = 64: 2 stages x 64 KB (128-byte rows);   KT = 32: NS stages x 32 KB (64-byte rows, three stages in flight at NS = 4).
// WT: W stored [K][N] (input gradients dX = dY W): its tile is two [KT][128] token-major images read with ds_read_b64_tr_b16.
template <int KT, int NS, bool WT, bool PLAIN>
__global__ __launch_bounds__(512) void gemm_big_kernel(GemmArgs a) {
    using T = bf16raw;
    constexpr int WM = 8, BIG = 256;
    constexpr int XB = BIG * (KT == 64 ? LDS_ROW : ROW64), WB = WT ? 2 * KT * 256 : XB, STAGE = XB + WB;
    constexpr int NIX = KT == 64 ? 4 : 2, NIW = WT ? KT / 16 : NIX, LPS = NIX + NIW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto xs = [&](int i) -> char* { return smem + i * STAGE; };
    auto ws = [&](int i) -> char* { return smem + i * STAGE + XB; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y > 1) {                                   // XCD-contiguous, banded tile order (see gemm_kernel)
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned w = xcd_contiguous(lin, total);
        const unsigned per_band = 8 * gridDim.y;
        const unsigned band = w / per_band, first = band * 8;
        const unsigned gsz = min(gridDim.x - first, 8u);
        bx = (int)(first + (w % per_band) % gsz);
        by = (int)((w % per_band) / gsz);
    }
    const int m_blk = bx * BIG, n_blk = by * BIG;
    const int wm0 = (wave & 1) * 128, wn0 = (wave >> 1) * 64;
    const int nsteps = a.K / KT;

    // per-thread LDS-DMA sources of stage 0 (a later stage adds its K offset) and destinations inside a stage
    const char* xsrc[NIX]; int xoff[NIX];
    const char* wsrc[NIW]; int woff[NIW];
    long wkstride = 0;
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
        if constexpr (KT == 64) {
            const int R0 = (wave * NIX + it) * 8, r = R0 + (lane >> 3), p = lane & 7;
            int gr = m_blk + r; if (gr >= a.M) gr = a.M - 1;
            xsrc[it] = a.X + (long)gr * a.ldx * 2 + ((p ^ (swz(r) >> 4)) << 4);
            xoff[it] = R0 * LDS_ROW;
        } else {
            const int R0 = (wave * NIX + it) * 16, r = R0 + (lane >> 2), p = lane & 3;
            int gr = m_blk + r; if (gr >= a.M) gr = a.M - 1;
            xsrc[it] = a.X + (long)gr * a.ldx * 2 + ((p ^ swz64(r)) << 4);
            xoff[it] = R0 * ROW64;
        }
    }
    if constexpr (!WT) {
#pragma unroll
        for (int it = 0; it < NIW; ++it) {
            if constexpr (KT == 64) {
                const int R0 = (wave * NIW + it) * 8, r = R0 + (lane >> 3), p = lane & 7;
                int gr = n_blk + r; if (gr >= a.N) gr = a.N - 1;
                wsrc[it] = a.W + (long)gr * a.ldw * 2 + ((p ^ (swz(r) >> 4)) << 4);
                woff[it] = R0 * LDS_ROW;
            } else {
                const int R0 = (wave * NIW + it) * 16, r = R0 + (lane >> 2), p = lane & 3;
                int gr = n_blk + r; if (gr >= a.N) gr = a.N - 1;
                wsrc[it] = a.W + (long)gr * a.ldw * 2 + ((p ^ swz64(r)) << 4);
                woff[it] = R0 * ROW64;
            }
        }
    } else {
        // waves 0-3 fill the [KT][128] image of columns n_blk .. +127, waves 4-7 that of n_blk + 128 .. +255 (4 token rows per instruction)
        const int half = wave >> 2, wq = wave & 3;
        wkstride = a.ldw * 2;
#pragma unroll
        for (int it = 0; it < NIW; ++it) {
            const int R0 = wq * (KT / 4) + it * 4, r = R0 + (lane >> 4), p = lane & 15;
            int col = n_blk + half * 128 + ((p ^ swz256(r)) << 3);
            if (col >= a.N) col = 0;
            wsrc[it] = a.W + ((long)r * a.ldw + col) * 2;
            woff[it] = half * KT * 256 + R0 * 256;
        }
    }
    auto issue = [&](int step, int buf) {
        char* xt = xs(buf); char* wt = ws(buf);
#pragma unroll
        for (int it = 0; it < NIX; ++it) glds16_asm(xsrc[it] + (long)step * KT * 2, xt + xoff[it]);
#pragma unroll
        for (int it = 0; it < NIW; ++it) glds16_asm(wsrc[it] + (WT ? (long)step * KT * wkstride : (long)step * KT * 2), wt + woff[it]);
    };
    f32x4 acc[4][WM];
    zero_acc(acc);
    const bool noload = a.dbg & 2;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nsteps && !noload) issue(p, p);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int younger = min(NS - 2, nsteps - 1 - s);
        if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
        else if (NS >= 4 && younger == NS - 3) wait_vmcnt<(NS >= 4 ? NS - 3 : 0) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nsteps && !noload) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const char* xt = xs(buf); const char* wt = ws(buf);
#pragma unroll
        for (int c = 0; c < KT / 32; ++c) {
            uint4 af[4], bfr[WM];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if constexpr (WT) af[m] = frag_tr256(wt + (wn0 >> 7) * KT * 256, (wn0 & 127) + 16 * m, c);
                else af[m] = KT == 64 ? frag_sw(wt, wn0 + 16 * m, c) : frag_sw64(wt, wn0 + 16 * m);
            }
#pragma unroll
            for (int n = 0; n < WM; ++n) bfr[n] = KT == 64 ? frag_sw(xt, wm0 + 16 * n, c) : frag_sw64(xt, wm0 + 16 * n);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    if ((a.dbg & 1) && acc[0][0][0] != 12345.678f) return;
    if (a.staged >= 0) {
        __syncthreads();                                   // every wave has left the ring: its space stages the output rows
        tile_epilogue_staged<WM>(a, acc, m_blk, n_blk, wm0, wn0, smem, a.staged);
        return;
    }
    tile_epilogue<T, WM, PLAIN>(a, acc, m_blk, n_blk, wm0, wn0, 0);
}

// ---- 128 x 256 tiles, 4 waves, TWO workgroups per CU ------------------------------------------------------------------------------
// The 256 x 256 kernel runs one workgroup per CU: with the operand loads removed it still spent half its time in the epilogue's
// stores (tools/big_gemm_bench.py, FW_GEMM_BIG_DBG: 62 us whole, 32 us without stores at 16384 x 1792 x 448), because nothing else
// runs on the CU while a tile drains.  Same wave tile here (128 rows x 64 columns, 32 accumulator tiles, 12 fragment reads per 32
// MFMAs), but the workgroup is 4 waves = 128 x 256 outputs on a 3-stage ring of 32-deep steps (72 KB): two workgroups share a CU, and
// one's store phase runs under the other's K loop.  X tile [128][64 B], W tile [256][64 B] (or two [32][128] token-major images).
template <int NS, bool WT, bool PLAIN>
__global__ __launch_bounds__(256, 2) void gemm_wide_kernel(GemmArgs a) {
    using T = bf16raw;
    constexpr int KT = 32, WM = 8, TM = 128, TN = 256;
    constexpr int XB = TM * ROW64, WB = WT ? 2 * KT * 256 : TN * ROW64, STAGE = XB + WB;
    constexpr int NIX = 2, NIW = 4, LPS = NIX + NIW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto xs = [&](int i) -> char* { return smem + i * STAGE; };
    auto ws = [&](int i) -> char* { return smem + i * STAGE + XB; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y > 1) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
        const unsigned w = xcd_contiguous(lin, total);
        const unsigned per_band = 8 * gridDim.y;
        const unsigned band = w / per_band, first = band * 8;
        const unsigned gsz = min(gridDim.x - first, 8u);
        bx = (int)(first + (w % per_band) % gsz);
        by = (int)((w % per_band) / gsz);
    }
    const int m_blk = bx * TM, n_blk = by * TN;
    const int wm0 = 0, wn0 = wave * 64;
    const int nsteps = a.K / KT;
    const char* xsrc[NIX]; int xoff[NIX];
    const char* wsrc[NIW]; int woff[NIW];
    long wkstride = 0;
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
        const int R0 = (wave * NIX + it) * 16, r = R0 + (lane >> 2), p = lane & 3;
        int gr = m_blk + r; if (gr >= a.M) gr = a.M - 1;
        xsrc[it] = a.X + (long)gr * a.ldx * 2 + ((p ^ swz64(r)) << 4);
        xoff[it] = R0 * ROW64;
    }
    if constexpr (!WT) {
#pragma unroll
        for (int it = 0; it < NIW; ++it) {
            const int R0 = (wave * NIW + it) * 16, r = R0 + (lane >> 2), p = lane & 3;
            int gr = n_blk + r; if (gr >= a.N) gr = a.N - 1;
            wsrc[it] = a.W + (long)gr * a.ldw * 2 + ((p ^ swz64(r)) << 4);
            woff[it] = R0 * ROW64;
        }
    } else {
        const int half = wave >> 1, wq = wave & 1;          // waves 0-1: columns n_blk .. +127, waves 2-3: the next 128; 16 token rows each
        wkstride = a.ldw * 2;
#pragma unroll
        for (int it = 0; it < NIW; ++it) {
            const int R0 = wq * 16 + it * 4, r = R0 + (lane >> 4), p = lane & 15;
            int col = n_blk + half * 128 + ((p ^ swz256(r)) << 3);
            if (col >= a.N) col = 0;
            wsrc[it] = a.W + ((long)r * a.ldw + col) * 2;
            woff[it] = half * KT * 256 + R0 * 256;
        }
    }
    auto issue = [&](int step, int buf) {
        char* xt = xs(buf); char* wt = ws(buf);
#pragma unroll
        for (int it = 0; it < NIX; ++it) glds16_asm(xsrc[it] + (long)step * KT * 2, xt + xoff[it]);
#pragma unroll
        for (int it = 0; it < NIW; ++it) glds16_asm(wsrc[it] + (WT ? (long)step * KT * wkstride : (long)step * KT * 2), wt + woff[it]);
    };
    f32x4 acc[4][WM];
    zero_acc(acc);
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nsteps) issue(p, p);
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int younger = min(NS - 2, nsteps - 1 - s);
        if (younger >= NS - 2) wait_vmcnt<(NS - 2) * LPS>();
        else if (NS >= 4 && younger == NS - 3) wait_vmcnt<(NS >= 4 ? NS - 3 : 0) * LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + NS - 1 < nsteps) issue(s + NS - 1, buf == 0 ? NS - 1 : buf - 1);
        const char* xt = xs(buf); const char* wt = ws(buf);
        uint4 af[4], bfr[WM];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (WT) af[m] = frag_tr256(wt + (wn0 >> 7) * KT * 256, (wn0 & 127) + 16 * m, 0);
            else af[m] = frag_sw64(wt, wn0 + 16 * m);
        }
#pragma unroll
        for (int n = 0; n < WM; ++n) bfr[n] = frag_sw64(xt, wm0 + 16 * n);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < WM; ++n) mma_chunk<T>(acc[m][n], af[m], bfr[n]);
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    if (a.staged >= 0) {
        __syncthreads();
        tile_epilogue_staged<WM>(a, acc, m_blk, n_blk, wm0, wn0, smem, a.staged);
        return;
    }
    tile_epilogue<T, WM, PLAIN>(a, acc, m_blk, n_blk, wm0, wn0, 0);
}
template <int NS, bool WT, bool PLAIN>
int launch_wide_p(const GemmArgs& a, hipStream_t st) {
    size_t lds = (size_t)NS * (128 * ROW64 + (WT ? 2 * 32 * 256 : 256 * ROW64));
    if (lds < (size_t)4 * 128 * EPI_LD) lds = (size_t)4 * 128 * EPI_LD;
    FW_SET_LDS_ONCE((gemm_wide_kernel<NS, WT, PLAIN>), lds);
    FW_KNAME("gemm_wide_kernel<%d,%s,%s>", NS, FW_B(WT), FW_B(PLAIN));
    hipLaunchKernelGGL((gemm_wide_kernel<NS, WT, PLAIN>), dim3(fw_cdiv(a.M, 128), fw_cdiv(a.N, 256), 1), dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}

template <int KT, int NS, bool WT, bool PLAIN>
int launch_big_p(const GemmArgs& a, hipStream_t st) {
    size_t lds = (size_t)NS * (256 * (KT == 64 ? LDS_ROW : ROW64) + (WT ? 2 * KT * 256 : 256 * (KT == 64 ? LDS_ROW : ROW64)));
    if (lds < (size_t)8 * 128 * EPI_LD) lds = (size_t)8 * 128 * EPI_LD;       // the staged epilogue's 8 x [128][136 B] regions
    FW_SET_LDS_ONCE((gemm_big_kernel<KT, NS, WT, PLAIN>), lds);
    FW_KNAME("gemm_big_kernel<%d,%d,%s,%s>", KT, NS, FW_B(WT), FW_B(PLAIN));
    hipLaunchKernelGGL((gemm_big_kernel<KT, NS, WT, PLAIN>), dim3(fw_cdiv(a.M, 256), fw_cdiv(a.N, 256), 1), dim3(512), lds, st, a);
    FW_LAUNCH_RET();
}
template <bool WT>
int launch_big(const GemmArgs& a, hipStream_t st) {
    // FW_GEMM_BIG: 0 off, 1 = 256 x 256, 64-deep steps / 2 stages, 2 = 32-deep / 4 stages, 3 = 32-deep / 3 stages;
    //              4 = 128 x 256 tiles at two workgroups per CU (gemm_wide_kernel, 3 stages), 5 = the same with 4 stages (one per CU)
    // default (mode 6): 128 x 256 tiles at two workgroups per CU, except the 65 536-wide encoder heads (N >= 16384: 1 024 tiles of
    // 256 x 256 -- 95 us against 106 us).  Measured per shape with tools/big_gemm_bench.py (gpurun_out/bgm*.txt of round 3).
    static const int mode_env = getenv("FW_GEMM_BIG") ? atoi(getenv("FW_GEMM_BIG")) : 6;
    const int mode = mode_env == 6 ? (a.N >= 16384 ? 1 : 4) : mode_env;
    const bool pl = plain_epilogue(a);
    if (mode == 4) return pl ? launch_wide_p<3, WT, true>(a, st) : launch_wide_p<3, WT, false>(a, st);
    if (mode == 5) return pl ? launch_wide_p<4, WT, true>(a, st) : launch_wide_p<4, WT, false>(a, st);
    if (mode == 2) return pl ? launch_big_p<32, 4, WT, true>(a, st) : launch_big_p<32, 4, WT, false>(a, st);
    if (mode == 3) return pl ? launch_big_p<32, 3, WT, true>(a, st) : launch_big_p<32, 3, WT, false>(a, st);
    return pl ? launch_big_p<64, 2, WT, true>(a, st) : launch_big_p<64, 2, WT, false>(a, st);
}

template <bool XT>
int launch_tr(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 4 * 64 * 256;
    FW_SET_LDS_ONCE(gemm_tr_kernel<XT>, lds);
    FW_KNAME("gemm_tr_kernel<%s>", FW_B(XT));
    hipLaunchKernelGGL(gemm_tr_kernel<XT>, dim3(fw_cdiv(a.M, 128), fw_cdiv(a.N, 128), a.splitk), dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}

// =====================================================================================================
// W-stationary streaming kernel for TALL-SKINNY products: M = tokens (10^4 .. 10^6), K * sizeof(T) <= 512 bytes.
// These are HBM-bound (C <= 224 stages of the U-Net: 45 .. 180 FLOP/B against a ridge of ~310), and the tiled
// kernel above spends its time in fill / barrier / drain of one or two K steps at 2 waves per SIMD.  Here
//   * a workgroup (8 waves) stages its W panel [bnp columns][K] ONCE into LDS (k-contiguous, transposing if needed),
//   * then every wave streams 32-row strips of X on its own: the MFMA B-fragments are loaded STRAIGHT from global
//     memory into registers (16 B per lane, k-contiguous rows -- no LDS, no barrier), X is read exactly once for
//     all columns of the panel, and the strip after next is requested before the epilogue of the current one.
// 2 workgroups = 16 independent waves per CU keep loads, MFMAs and stores of different strips in flight together.
#define STREAM_MTS(NCH, EXT) 2          /* 64-row strips (4) for the K <= 64 plain variant measured no gain: 71.7 vs 77.5 us at 786 432 x 88 x 28, slower elsewhere */
template <typename T, int NCH, bool WT, int EXT>        // EXT: 0 no row-dependent epilogue operand, 1 GELU' input (T), 2 f32 residual
__global__ __launch_bounds__(512, (NCH == 2 && !EXT) ? 4 : 2) void gemm_stream_kernel(GemmArgs a, int bnp) {   // 2nd argument: waves per SIMD (2 workgroups per CU = 4)
    constexpr int SZ = TT<T>::SZ, E = TT<T>::E16;
    constexpr int RL = NCH * 64;                          // bytes of K per LDS row (NCH even: whole 128-byte swizzle groups)
    // rows per strip = 16 * MTS.  A wave has ONE strip's loads in flight: 32-row strips of a K <= 128 operand are 2-8 KB per wave, and with
    // the stores removed the kernel still read at only 2.2 TB/s (tools/stream_gemm_bench.py) -- 64-row strips double the bytes in flight
    constexpr int MTS = STREAM_MTS(NCH, EXT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = lane_id(), wave = threadIdx.x >> 6;
    const int n_blk = blockIdx.y * bnp;
    const int kbytes = a.K * SZ;

    // ---- stage the W panel: LDS row = output column n, zero past K and past N -----------------------
    if constexpr (!WT) {
        constexpr int SLOTS = RL / 16;
        for (int idx = threadIdx.x; idx < bnp * SLOTS; idx += 512) {
            const int row = idx / SLOTS, kb = (idx % SLOTS) * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n_blk + row < a.N && kb < kbytes) {
                v = *reinterpret_cast<const uint4*>(a.W + (long)(n_blk + row) * a.ldw * SZ + kb);
                const int valid = kbytes - kb;
                if (valid < 16) {
                    if (valid <= 12) v.w = 0;
                    if (valid <= 8) v.z = 0;
                    if (valid <= 4) v.y = 0;
                }
            }
            *reinterpret_cast<uint4*>(smem + row * RL + (kb ^ swz(row))) = v;
        }
    } else {
        // W is [K][N] (n contiguous): a thread takes 4 k-rows x E columns and writes E rows x 4 k (8 or 16 bytes)
        constexpr int KQ = RL / (4 * SZ);
        const int nblocks = bnp / E;
        for (int idx = threadIdx.x; idx < KQ * nblocks; idx += 512) {
            const int ib = idx % nblocks, kq = idx / nblocks;
            uint4 r[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = kq * 4 + kk;
                r[kk] = make_uint4(0, 0, 0, 0);
                if (k < a.K && n_blk + ib * E < a.N) r[kk] = *reinterpret_cast<const uint4*>(a.W + ((long)k * a.ldw + n_blk + ib * E) * SZ);
            }
            if constexpr (sizeof(T) == 4) {
                const int r0 = ib * 4, cb = kq * 16;
                *reinterpret_cast<uint4*>(smem + (r0 + 0) * RL + (cb ^ swz(r0 + 0))) = make_uint4(r[0].x, r[1].x, r[2].x, r[3].x);
                *reinterpret_cast<uint4*>(smem + (r0 + 1) * RL + (cb ^ swz(r0 + 1))) = make_uint4(r[0].y, r[1].y, r[2].y, r[3].y);
                *reinterpret_cast<uint4*>(smem + (r0 + 2) * RL + (cb ^ swz(r0 + 2))) = make_uint4(r[0].z, r[1].z, r[2].z, r[3].z);
                *reinterpret_cast<uint4*>(smem + (r0 + 3) * RL + (cb ^ swz(r0 + 3))) = make_uint4(r[0].w, r[1].w, r[2].w, r[3].w);
            } else {
                const int r0 = ib * 8, cb = kq * 8;
                const unsigned w0[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
                const unsigned w1[4] = {r[1].x, r[1].y, r[1].z, r[1].w};
                const unsigned w2[4] = {r[2].x, r[2].y, r[2].z, r[2].w};
                const unsigned w3[4] = {r[3].x, r[3].y, r[3].z, r[3].w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint2 lo = make_uint2(__builtin_amdgcn_perm(w1[d], w0[d], 0x05040100u), __builtin_amdgcn_perm(w3[d], w2[d], 0x05040100u));
                    const uint2 hi = make_uint2(__builtin_amdgcn_perm(w1[d], w0[d], 0x07060302u), __builtin_amdgcn_perm(w3[d], w2[d], 0x07060302u));
                    *reinterpret_cast<uint2*>(smem + (r0 + 2 * d) * RL + (cb ^ swz(r0 + 2 * d))) = lo;
                    *reinterpret_cast<uint2*>(smem + (r0 + 2 * d + 1) * RL + (cb ^ swz(r0 + 2 * d + 1))) = hi;
                }
            }
        }
    }
    {                                                     // bias of the panel's columns (zeros when there is none / past N)
        float* sb = reinterpret_cast<float*>(smem + (size_t)bnp * RL);
        for (int i = threadIdx.x; i < bnp; i += 512) sb[i] = (a.bias && n_blk + i < a.N) ? a.bias[n_blk + i] : 0.f;
    }
    __syncthreads();                                      // the only barrier: from here on the waves run independently

    const int strips = (a.M + 16 * MTS - 1) / (16 * MTS);
    const int stride = gridDim.x * 8;
    const int ncols = min(bnp, a.N - n_blk);
    const int nnb = (ncols + 63) >> 6;
    uint4 xf[MTS][NCH];
    auto issue_x = [&](int strip) {                       // phase 1: the loads only (clamped coordinates), see StageDirect::load
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) {
            const int row = strip * (16 * MTS) + mt * 16 + (l & 15);
            const char* src = a.X + (long)(row < a.M ? row : a.M - 1) * a.ldx * SZ;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int kb = c * 64 + ((l >> 4) << 4);
                xf[mt][c] = *reinterpret_cast<const uint4*>(src + (kb < kbytes ? kb : 0));
            }
        }
    };
    auto mask_x = [&](int strip) {                        // phase 2: zero what lies past M / K
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) {
            const bool rok = strip * (16 * MTS) + mt * 16 + (l & 15) < a.M;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int valid = rok ? kbytes - (c * 64 + ((l >> 4) << 4)) : 0;
                uint4 v = xf[mt][c];
                v.x = valid > 0 ? v.x : 0u; v.y = valid > 4 ? v.y : 0u; v.z = valid > 8 ? v.z : 0u; v.w = valid > 12 ? v.w : 0u;
                xf[mt][c] = v;
            }
        }
    };
    // vmcnt counts loads AND stores in issue order: a global load issued after stores cannot be waited for without waiting for
    // those stores to be acknowledged (1-2 us).  The bias therefore comes from LDS (staged with the panel): without a row-dependent
    // operand (EXT) the epilogue issues no global load at all and its stores never block the wave.
    const float* sbias = reinterpret_cast<const float*>(smem + (size_t)bnp * RL);
    int strip = blockIdx.x * 8 + wave;
    if (strip < strips) issue_x(strip);
    for (; strip < strips; strip += stride) {
        mask_x(strip);
        const int m_lane = strip * (16 * MTS) + (l & 15);
        float rs[MTS];
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) {
            const int m = m_lane + mt * 16;
            rs[mt] = a.rowscale ? a.rowscale[(m < a.M ? m : a.M - 1) / a.rows_per_scale] : 1.0f;
        }
        if constexpr (EXT == 1 && NCH <= 4) {
            // GELU' input of the fc2 input gradients: its loads for column block nb + 1 are issued BEFORE block nb is applied and
            // stored (two register sets, two blocks per loop trip).  Fetched and consumed inside the same block, every (block,
            // 16-row group) exposed one full memory latency -- 14 per strip at N = 448 -- and the waves of this variant sat in
            // s_waitcnt for 3/4 of their cycles (SQ_WAIT_ANY 0.75).  vmcnt retires in issue order, so waiting for these loads
            // never waits for the stores issued after them.  (NCH = 8 has no registers left for the second set.)
            constexpr int EW = sizeof(T) == 2 ? 1 : 2;                       // uint2 words per 4 elements of T
            struct ExtSet { uint2 v[MTS][4][EW]; };
            auto fetch_ext = [&](ExtSet& e, int nb) {
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt) {
                    const int m = m_lane + mt * 16;
                    const T* ap = reinterpret_cast<const T*>(a.aux) + (long)(m < a.M ? m : a.M - 1) * a.ldaux;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n0 = n_blk + nb * 64 + i * 16 + ((l >> 4) << 2);
                        const uint2* src = reinterpret_cast<const uint2*>(ap + (n0 < a.N ? n0 : 0));
#pragma unroll
                        for (int q = 0; q < EW; ++q) e.v[mt][i][q] = src[q];
                    }
                }
            };
            auto block = [&](int nb, const ExtSet& cur, ExtSet& nxt) {
                f32x4 acc[4][MTS];
                zero_acc(acc);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    uint4 af[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = nb * 64 + i * 16 + (l & 15);
                        af[i] = *reinterpret_cast<const uint4*>(smem + row * RL + ((c * 64 + ((l >> 4) << 4)) ^ swz(row)));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int mt = 0; mt < MTS; ++mt) mma_chunk<T>(acc[i][mt], af[i], xf[mt][c]);
                    }
                }
                if (nb + 1 < nnb) fetch_ext(nxt, nb + 1);
                if (nb == nnb - 1 && strip + stride < strips) issue_x(strip + stride);
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt) {
                    const int m = m_lane + mt * 16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n0 = n_blk + nb * 64 + i * 16 + ((l >> 4) << 2);
                        const uint4 ex = EW == 1 ? make_uint4(cur.v[mt][i][0].x, cur.v[mt][i][0].y, 0u, 0u)
                                                 : make_uint4(cur.v[mt][i][0].x, cur.v[mt][i][0].y, cur.v[mt][i][EW - 1].x, cur.v[mt][i][EW - 1].y);
                        if (m < a.M && n0 < a.N) epi_apply<T>(a, *reinterpret_cast<const f32x4*>(sbias + (n0 - n_blk)), ex, acc[i][mt], m, n0, rs[mt], 0);
                    }
                }
            };
            ExtSet ea, eb;
            fetch_ext(ea, 0);
#pragma unroll 1
            for (int nb = 0; nb < nnb; nb += 2) {
                block(nb, ea, eb);
                if (nb + 1 < nnb) block(nb + 1, eb, ea);
            }
        } else {
#pragma unroll 1
            for (int nb = 0; nb < nnb; ++nb) {
                f32x4 acc[4][MTS];
                zero_acc(acc);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    uint4 af[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = nb * 64 + i * 16 + (l & 15);
                        af[i] = *reinterpret_cast<const uint4*>(smem + row * RL + ((c * 64 + ((l >> 4) << 4)) ^ swz(row)));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int mt = 0; mt < MTS; ++mt) mma_chunk<T>(acc[i][mt], af[i], xf[mt][c]);
                    }
                }
                if (nb == nnb - 1 && strip + stride < strips) issue_x(strip + stride);   // request the next strip, then write this one out
                if constexpr (EXT == 0 && sizeof(T) == 2) {
                    if (a.staged == 0) {
                        // bf16 output, bias only: the 16 x 64 sub-tiles go through the wave's own LDS region and leave as whole 128-byte
                        // row segments (16 bytes per lane, 8 rows per instruction) instead of 8-byte pieces of 16 rows -- with the stores
                        // removed this kernel runs 3.4x faster (22.6 vs 76.4 us at 786 432 x 88 x 28: tools/stream_gemm_bench.py), i.e. it
                        // is bound by how its output reaches HBM, not by its loads
                        char* mine = smem + (size_t)bnp * RL + (size_t)bnp * 4 + wave * (16 * EPI_LD);
                        const int rsub = l >> 3, cb = (l & 7) * 16;
                        const int ncol = n_blk + nb * 64 + (l & 7) * 8;
#pragma unroll
                        for (int mt = 0; mt < MTS; ++mt) {
                            char* rowp = mine + (l & 15) * EPI_LD + ((l >> 4) << 3);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const f32x4 v = acc[i][mt] + *reinterpret_cast<const f32x4*>(sbias + nb * 64 + i * 16 + ((l >> 4) << 2));
                                *reinterpret_cast<uint2*>(rowp + i * 32) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                            for (int it = 0; it < 2; ++it) {
                                const int r = it * 8 + rsub, m = strip * (16 * MTS) + mt * 16 + r;
                                const uint4 v = *reinterpret_cast<const uint4*>(mine + r * EPI_LD + cb);
                                if ((a.dbg & 1) && v.x != 0x12345678u) continue;
                                if (m < a.M && ncol < a.N) *reinterpret_cast<uint4*>(a.C + ((long)m * a.ldc + ncol) * 2) = v;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the region is rewritten by the next 16 rows
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        }
                        continue;
                    }
                }
                // a row-dependent operand (EXT): the 4 quads of 16 rows are fetched together before they are applied; otherwise quad
                // by quad, which keeps the kernel at 4 waves per SIMD
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt) {
                    const int m = m_lane + mt * 16;
                    const int mc = m < a.M ? m : a.M - 1;
                    uint4 ext[4];
                    if constexpr (EXT != 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int n0 = n_blk + nb * 64 + i * 16 + ((l >> 4) << 2);
                            epi_fetch<T>(a, ext[i], mc, n0 < a.N ? n0 : 0, 0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int n0 = n_blk + nb * 64 + i * 16 + ((l >> 4) << 2);
                        if ((a.dbg & 1) && acc[i][mt][0] != 12345.678f) continue;          // measurement only: no stores
                        if (m < a.M && n0 < a.N) epi_apply<T>(a, *reinterpret_cast<const f32x4*>(sbias + (n0 - n_blk)), ext[i], acc[i][mt], m, n0, rs[mt], 0);
                    }
                }
            }
        }
    }
}

template <typename T, int NCH, bool WT, int EXT>
int launch_stream(const GemmArgs& a, hipStream_t st) {
    constexpr int RL = NCH * 64;
    // staged bf16 epilogue (EXT == 0): 8 regions of [16 rows][136 B] behind the panel; two workgroups per CU share 160 KB
    const bool stg = EXT == 0 && sizeof(T) == 2 && a.staged == 0;
    const size_t stage_b = stg ? (size_t)8 * 16 * EPI_LD : 0;
    const int maxb = (int)(((stg ? 60 * 1024 : 64 * 1024)) / RL) & ~63;   // panel columns that fit (2 workgroups per CU)
    const int ny = fw_cdiv(a.N, maxb);
    const int bnp = fw_cdiv(fw_cdiv(a.N, ny), 64) * 64;
    const size_t lds = (size_t)bnp * RL + (size_t)bnp * 4 + stage_b;       // W panel + bias of its columns (+ the staging regions)
    FW_SET_LDS_ONCE((gemm_stream_kernel<T, NCH, WT, EXT>), 80 * 1024);
    const int strips = fw_cdiv(a.M, 16 * STREAM_MTS(NCH, EXT));
    int gx = fw_cdiv(strips, 8);
    const int cap = 512 / ny > 0 ? 512 / ny : 1;                    // 256 CUs x 2 workgroups
    if (gx > cap) gx = cap;
    FW_KNAME("gemm_stream_kernel<%s,%d,%s,%d>", tname<T>(), NCH, FW_B(WT), EXT);
    hipLaunchKernelGGL((gemm_stream_kernel<T, NCH, WT, EXT>), dim3(gx, ny), dim3(512), lds, st, a, bnp);
    FW_LAUNCH_RET();
}

template <typename T, int NCH>
int dispatch_stream_n(const GemmArgs& a, int wt, hipStream_t st) {
    // a row-dependent epilogue operand is prefetched one column block ahead (more registers): 1 = GELU' input, 2 = f32 residual
    const int ext = a.act == 2 ? 1 : (a.residual != nullptr ? 2 : 0);
    if (wt) return ext == 1 ? launch_stream<T, NCH, true, 1>(a, st) : ext == 2 ? launch_stream<T, NCH, true, 2>(a, st) : launch_stream<T, NCH, true, 0>(a, st);
    return ext == 1 ? launch_stream<T, NCH, false, 1>(a, st) : ext == 2 ? launch_stream<T, NCH, false, 2>(a, st) : launch_stream<T, NCH, false, 0>(a, st);
}
template <typename T>
int dispatch_stream(const GemmArgs& a, int wt, hipStream_t st) {
    const int nch = fw_cdiv(a.K * TT<T>::SZ, 128) * 2;
    if (nch <= 2) return dispatch_stream_n<T, 2>(a, wt, st);
    if (nch <= 4) return dispatch_stream_n<T, 4>(a, wt, st);
    return dispatch_stream_n<T, 8>(a, wt, st);
}

template <typename T, int BN, bool XT, bool WT, bool GX, bool GW>
int launch(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 2 * (size_t)(BM + BN) * LDS_ROW;
    FW_SET_LDS_ONCE((gemm_kernel<T, BN, XT, WT, GX, GW>), lds);
    dim3 grid(fw_cdiv(a.M, BM), fw_cdiv(a.N, BN), a.splitk);
    FW_KNAME("gemm_kernel<%s,%d,%s,%s,%s,%s>", tname<T>(), BN, FW_B(XT), FW_B(WT), FW_B(GX), FW_B(GW));
    hipLaunchKernelGGL((gemm_kernel<T, BN, XT, WT, GX, GW>), grid, dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}

template <typename T, int BN>
int dispatch_trans(const GemmArgs& a, int xt, int wt, hipStream_t st) {
    // direct global->LDS staging for k-contiguous operands whose K range is a whole number of 128-byte steps
    const int kt = 128 / TT<T>::SZ;
    const bool whole = a.K % kt == 0;
    const bool gx = !xt && whole && a.x_op == 0, gw = !wt && whole && a.w_op == 0;
    if (!xt && !wt) {
        // 64-byte K steps: for K that is not a whole number of 128-byte steps (K = 224, 672 in bf16: the round-1 tile kernel took
        // those with register staging: 28.5 -> 19.5 us at 16384 x 672 x 224).  On whole-step shapes it measured 5-10 % SLOWER than the
        // two-stage 128-byte ring at BN = 128 (FW_GEMM_RING64=2 forces it everywhere, 0 disables it).
        static const int ring64 = getenv("FW_GEMM_RING64") ? atoi(getenv("FW_GEMM_RING64")) : 1;
        const int kt64 = 64 / TT<T>::SZ;
        if (ring64 && a.x_op == 0 && a.w_op == 0 && a.K % kt64 == 0 && a.kper % kt64 == 0 && (ring64 == 2 || (!whole && BN == 128)))
        {
            // three 16 KB stages = 48 KB: THREE workgroups per CU (160 VGPRs allow it) -- 19.4 -> 17.3 us against four stages at two
            static const int ns64 = getenv("FW_GEMM_RING64_NS") ? atoi(getenv("FW_GEMM_RING64_NS")) : 3;
            if (ns64 == 4) return launch_ring64<T, BN, 4>(a, st);
            return launch_ring64<T, BN, 3>(a, st);
        }
        static const int ring = getenv("FW_GEMM_RING") ? atoi(getenv("FW_GEMM_RING")) : 1;        // 0: gemm_kernel (one stage in flight)
        if (gx && gw && ring && a.kper % kt == 0) {
            // LDS per workgroup decides the residency: 2 stages of 128 x 128 = 64 KB -> 2 workgroups per CU (ring 1, default);
            // deeper rings of one resident workgroup measured SLOWER (tools/probe/glds_probe.hip: 28 -> 42 us)
            if (ring == 2) return BN == 128 ? launch_ring<T, BN, 3>(a, st) : launch_ring<T, BN, 3>(a, st);
            if (ring == 3) return BN == 128 ? launch_ring<T, BN, 4>(a, st) : launch_ring<T, BN, 4>(a, st);
            // BN = 64: 3 x 24 KB = 72 KB still leaves two workgroups per CU -- pays on long K loops (3072 x 448 x 3584: 48.6 -> 34.2 us,
            // 1024 x 896 x 7168: 88.7 -> 62.1), costs 10-20 % on 7-step ones
            if ((ring == 1 || ring == 4) && BN == 64 && a.kper >= 28 * kt) return launch_ring<T, BN, 3>(a, st);
            return launch_ring<T, BN, 2>(a, st);
        }
        if (gx && gw) return launch<T, BN, false, false, true, true>(a, st);
        if (gx) return launch<T, BN, false, false, true, false>(a, st);
        if (gw) return launch<T, BN, false, false, false, true>(a, st);
        return launch<T, BN, false, false, false, false>(a, st);
    }
    if (!xt && wt) return gx ? launch<T, BN, false, true, true, false>(a, st) : launch<T, BN, false, true, false, false>(a, st);
    if (xt && !wt) return gw ? launch<T, BN, true, false, false, true>(a, st) : launch<T, BN, true, false, false, false>(a, st);
    return launch<T, BN, true, true, false, false>(a, st);
}

}  // namespace

// C-ABI.  Declared in include/fwair.h.
// Which kernel the last fw_gemm call of this thread was dispatched to (measurement aid: bench.py labels its per-launch timings
// with it so that they line up with the rocprofv3 kernel names): family * 100000 + BN * 100 + xT * 10 + wT,
// family 0 = gemm_kernel (BN = 64 / 128), 1 = gemm_tr_kernel, 2 = gemm_stream_kernel.
static thread_local int g_last_variant = 0;
extern "C" int fw_gemm_last_variant(void) { return g_last_variant; }
extern "C" int fw_gemm_last_kernel(char* buf, int n) {
    if (!buf || n <= 0) return -1;
    snprintf(buf, (size_t)n, "%s", g_last_kernel);
    return (int)strlen(g_last_kernel);
}

// Grouped weight gradients (see gemm_wgrad_group_kernel).  tab: device int64 [nprob][16] =
//   {X (dY, bf16 [tokens][ldx]), W (x, bf16 [tokens][ldw]), C (f32), ldx, ldw, ldc, M (rows of dW), N (columns), K (tokens), kper (tokens
//    per slice, multiple of 32), splitk, xsum (f32 or 0), c_zstride, xsum_zstride, accumulate, 0};
// probs: device scratch of nprob * fw_wgrad_group_prob_bytes() bytes the kernel's argument blocks are built in;
// items: device int32 [nitems][4] = {problem, m tile, n tile, slice}.  splitk == 1: the tile ADDS into C (accumulate = 1) or stores;
// splitk > 1: slice z stores its partial tile at C + z * c_zstride (the caller folds the slabs, fw_slab_reduce_multi).
extern "C" int fw_wgrad_group_prob_bytes(void) { return (int)sizeof(GemmArgs); }
namespace {
__global__ void wgrad_group_fill_kernel(const long long* __restrict__ tab, GemmArgs* __restrict__ probs, int nprob) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nprob) return;
    const long long* t = tab + (size_t)i * 16;
    GemmArgs a;
    a.X = reinterpret_cast<const char*>(t[0]); a.W = reinterpret_cast<const char*>(t[1]); a.C = reinterpret_cast<char*>(t[2]);
    a.ldx = t[3]; a.ldw = t[4]; a.ldc = t[5]; a.M = (int)t[6]; a.N = (int)t[7]; a.K = (int)t[8]; a.kper = (int)t[9]; a.splitk = (int)t[10];
    a.x_op = a.w_op = 0; a.bias = nullptr; a.act = 0; a.slope = 0.f; a.aux = nullptr; a.ldaux = 0; a.rowscale = nullptr; a.rows_per_scale = 1;
    a.residual = nullptr; a.ldr = 0; a.out_f32 = 1; a.accumulate = (int)t[14]; a.C2 = nullptr; a.ldc2 = 0;
    a.xsum = reinterpret_cast<float*>(t[11]); a.c_zstride = t[12]; a.xsum_zstride = t[13]; a.alpha = 1.0f; a.staged = -1; a.dbg = 0;
    probs[i] = a;
}
}  // namespace
extern "C" int fw_wgrad_group(const void* tab, void* probs, int nprob, const void* items, int nitems, int tile, void* stream) {
    FW_CHECK_ARG(tab && probs && items && nprob > 0 && nitems > 0 && (tile == 128 || tile == 256));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_group_fill_kernel, dim3((nprob + 63) / 64), dim3(64), 0, st, (const long long*)tab, (GemmArgs*)probs, nprob);
    if (tile == 128) {
        const size_t lds = (size_t)3 * (32 * 256 + 32 * 256);
        FW_SET_LDS_ONCE(gemm_wgrad_group_kernel, lds);
        hipLaunchKernelGGL(gemm_wgrad_group_kernel, dim3(nitems), dim3(256), lds, st, (const GemmArgs*)probs, (const int4*)items);
    } else {
        const size_t lds = (size_t)3 * 4 * 32 * 256;
        FW_SET_LDS_ONCE(gemm_wgrad_group_big_kernel, lds);
        hipLaunchKernelGGL(gemm_wgrad_group_big_kernel, dim3(nitems), dim3(512), lds, st, (const GemmArgs*)probs, (const int4*)items);
    }
    FW_LAUNCH_RET();
}

extern "C" int fw_gemm(int dtype, const void* X, long ldx, int x_trans, int x_op, const void* W, long ldw,
                       int w_trans, int w_op, void* C, long ldc, int out_f32, int accumulate, int M, int N,
                       int K, float alpha, const float* bias, int act, float slope, const void* aux,
                       long ldaux, const float* rowscale, int rows_per_scale, const float* residual,
                       long ldr, int splitk, void* C2, long ldc2, float* xsum, long c_zstride, long xsum_zstride,
                       void* stream) {
    const int sz = dtype == FW_DT_BF16 ? 2 : 4;
    const int e16 = 16 / sz;
    FW_CHECK_ARG(dtype == FW_DT_F32 || dtype == FW_DT_BF16);
    FW_CHECK_ARG(M > 0 && N > 0 && K > 0 && X && W && C);
    FW_CHECK_ARG(N % 4 == 0);
    FW_CHECK_ARG(ldx % e16 == 0 && ldw % e16 == 0 && ldc % 4 == 0);
    const int csz = (out_f32 || dtype == FW_DT_F32) ? 4 : 2;
    FW_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & (4 * csz - 1)) == 0);
    if (!x_trans) FW_CHECK_ARG((K * sz) % 4 == 0 && ldx >= K); else FW_CHECK_ARG(ldx >= M);
    if (!w_trans) FW_CHECK_ARG((K * sz) % 4 == 0 && ldw >= K); else FW_CHECK_ARG(ldw >= N);
    FW_CHECK_ARG(!accumulate || out_f32 || dtype == FW_DT_F32);
    FW_CHECK_ARG(splitk >= 1 && (splitk == 1 || accumulate || c_zstride > 0));
    FW_CHECK_ARG(c_zstride == 0 || (out_f32 && c_zstride % 4 == 0));
    FW_CHECK_ARG(act >= 0 && act <= 3 && (act != 2 || aux));
    FW_CHECK_ARG(!rowscale || rows_per_scale > 0);
    if (bias) FW_CHECK_ARG(((uintptr_t)bias & 15) == 0);
    if (residual) FW_CHECK_ARG(((uintptr_t)residual & 15) == 0 && ldr % 4 == 0);
    const int kt = 128 / sz;
    GemmArgs a;
    a.X = (const char*)X; a.W = (const char*)W; a.C = (char*)C;
    a.ldx = ldx; a.ldw = ldw; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.x_op = x_op; a.w_op = w_op; a.bias = bias; a.act = act; a.slope = slope;
    a.aux = (const char*)aux; a.ldaux = ldaux; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale;
    a.residual = residual; a.ldr = ldr;
    a.out_f32 = (out_f32 || dtype == FW_DT_F32) ? 1 : 0;
    a.accumulate = accumulate;
    a.splitk = splitk;
    a.C2 = (char*)C2; a.ldc2 = ldc2; a.xsum = xsum; a.c_zstride = c_zstride; a.xsum_zstride = xsum_zstride;
    FW_CHECK_ARG(!xsum || x_trans);
    FW_CHECK_ARG(!C2 || (((uintptr_t)C2 & (4 * sz - 1)) == 0 && ldc2 % 4 == 0));
    a.kper = fw_cdiv(fw_cdiv(K, kt), splitk) * kt;
    a.alpha = alpha;
    static const int use_staged = getenv("FW_GEMM_STAGED") ? atoi(getenv("FW_GEMM_STAGED")) : 1;
    a.staged = use_staged ? staged_mode(a) : -1;
    static const int dbg = getenv("FW_GEMM_BIG_DBG") ? atoi(getenv("FW_GEMM_BIG_DBG")) : 0;
    a.dbg = dbg;
    hipStream_t st = (hipStream_t)stream;
    // tall-skinny products stream X past a W panel held in LDS (gemm_stream_kernel)
    static const long stream_min_m = getenv("FW_GEMM_STREAM_MIN_M") ? atol(getenv("FW_GEMM_STREAM_MIN_M")) : 32768;
    if (!x_trans && splitk == 1 && !accumulate && x_op == 0 && w_op == 0 && !xsum && K * sz <= 512 && M >= stream_min_m) {
        g_last_variant = 200000 + (w_trans ? 1 : 0);
        return dtype == FW_DT_BF16 ? dispatch_stream<bf16raw>(a, w_trans, st) : dispatch_stream<float>(a, w_trans, st);
    }
    // compute-bound shapes on 256 x 256 tiles: at least 160 tiles (most of the 256 CUs busy in the only or last round)
    static const int big = getenv("FW_GEMM_BIG") ? atoi(getenv("FW_GEMM_BIG")) : 6;
    static const long big_min_tiles = getenv("FW_GEMM_BIG_MIN_TILES") ? atol(getenv("FW_GEMM_BIG_MIN_TILES")) : 160;
    if (big && dtype == FW_DT_BF16 && !x_trans && x_op == 0 && w_op == 0 && !xsum && splitk == 1 && !accumulate && K % 64 == 0 && K >= 256 &&
        ldx % 8 == 0 && ldw % 8 == 0 && (long)fw_cdiv(M, 256) * fw_cdiv(N, 256) >= big_min_tiles) {
        g_last_variant = 300000 + (w_trans ? 1 : 0);
        return w_trans ? launch_big<true>(a, st) : launch_big<false>(a, st);
    }
    // bf16 products whose W is stored [K][N] (weight gradients: X token-major too; input gradients: X k-contiguous), whole
    // 64-deep K steps: W (and X) tiles go to LDS as they are and are read with transposing LDS reads (gemm_tr_kernel)
    static const int use_tr = getenv("FW_GEMM_TR") ? atoi(getenv("FW_GEMM_TR")) : 3;
    // weight gradients with N <= 64 (the C = 28 / 56 stages: 56 x 28 x 786432, ...) on the ring kernel too: its 128 x 128 tile is mostly
    // masked there, but the long reduction is what costs -- 62.3 -> 48.2, 57.3 -> 34.7, 28.2 -> 17.4 us against the 128 x 64 tile kernel
    static const int tr_small_n = getenv("FW_GEMM_TR_SMALL_N") ? atoi(getenv("FW_GEMM_TR_SMALL_N")) : 1;
    if (dtype == FW_DT_BF16 && w_trans && x_op == 0 && w_op == 0 && (N > 64 || (tr_small_n && x_trans && N >= 8)) && K % 64 == 0 && a.kper % 64 == 0 && ldw % 8 == 0) {
        static const int ring = getenv("FW_GEMM_TR_RING") ? atoi(getenv("FW_GEMM_TR_RING")) : 1;     // 0: one stage in flight (gemm_tr_kernel)
        if (x_trans && (use_tr & 1) && ldx % 8 == 0) {
            g_last_variant = 100011;
            // three 16 KB stages and a 3-waves-per-SIMD register budget (141 VGPRs, no spills): THREE workgroups per CU.  265.6 -> 266.9
            // images/s against four stages at two workgroups (ring == 6 keeps that form); 155 -> 138 us at 65536 x 448 x 1024
            if (ring == 1) return launch_tr_ring<true, 32, 3>(a, st);
            if (ring == 6) return launch_tr_ring<true, 32, 4>(a, st);
            if (ring == 5) return launch_tr_ring<true, 64, 2>(a, st);
            if (ring == 2) return launch_tr_ring<true, 64, 3>(a, st);
            if (ring == 3) return launch_tr_ring<true, 64, 4>(a, st);
            if (ring == 4) return launch_tr_ring<true, 32, 5>(a, st);
            return launch_tr<true>(a, st);
        }
        static const long tr_min_tiles = getenv("FW_GEMM_TR_MIN_TILES") ? atol(getenv("FW_GEMM_TR_MIN_TILES")) : 200;   // 200..383 tiles: 77 -> 47 us at 4096 x 896 x 3584; below 200 the old kernel's 128 x 64 tiles fill more CUs
        if (!x_trans && (use_tr & 2) && !xsum && (long)fw_cdiv(M, 128) * fw_cdiv(N, 128) * splitk >= tr_min_tiles) {
            g_last_variant = 100001;
            // plain store and MORE blocks than CUs: the 32-deep form (48 KB, 126 VGPRs: three workgroups per CU) -- co-residency pays
            // once a CU has more than one block to run (16384 x 448 x 1792: 53.1 -> 45.1 us, 512 blocks); with at most one block per
            // CU the 64-deep three-stage ring stays ahead (4096 x 896 x 3584: 49.3 vs 54.4 us, 224 blocks).  0: never, 2: always
            static const int dx32 = getenv("FW_GEMM_TR_DX32") ? atoi(getenv("FW_GEMM_TR_DX32")) : 1;
            if (dx32 && plain_epilogue(a) && (dx32 == 2 || (long)fw_cdiv(M, 128) * fw_cdiv(N, 128) * splitk > 256))
                return launch_tr_ring<false, 32, 3>(a, st);
            if (ring == 5) return launch_tr_ring<false, 64, 2>(a, st);
            // 3 stages = 96 KB = ONE workgroup per CU: fastest while the epilogue is a plain store.  An epilogue with an operand of its
            // own (GELU' input) keeps the CU's only 4 waves off the MFMAs for as long as the K loop took; 2 stages = 64 KB lets a
            // second workgroup's K loop run under it: 135 -> 105 us at 16384 x 1792 x 448, 101 -> 71 us at 4096 x 3584 x 896.
            static const int np2 = getenv("FW_GEMM_TR_NP2") ? atoi(getenv("FW_GEMM_TR_NP2")) : 1;
            if (ring == 1 && np2 && !plain_epilogue(a)) return launch_tr_ring<false, 64, 2>(a, st);
            if (ring == 2 || ring == 1) return launch_tr_ring<false, 64, 3>(a, st);
            if (ring == 3 || ring == 4) return launch_tr_ring<false, 64, 4>(a, st);
            return launch_tr<false>(a, st);
        }
    }
    // the same input-gradient product with K a multiple of 32 only (K = 224, 336 is not): 32-deep steps, X in 64-byte rows
    if (dtype == FW_DT_BF16 && w_trans && !x_trans && x_op == 0 && w_op == 0 && N > 64 && K % 64 != 0 && K % 32 == 0 && ldw % 8 == 0 && !xsum
        && splitk == 1 && (long)fw_cdiv(M, 128) * fw_cdiv(N, 128) >= 200) {
        static const int r32 = getenv("FW_GEMM_TR_K32") ? atoi(getenv("FW_GEMM_TR_K32")) : 2;     // 1: plain epilogues only
        if (r32 && (r32 == 2 || plain_epilogue(a))) {          // with the lean GELU' epilogue: 44.4 -> 34.2 us at 16384 x 896 x 224 against the round-1 tile kernel
            g_last_variant = 100001;
            return launch_tr_ring<false, 32, 3>(a, st);
        }
    }
    // 128x64 tiles when N is narrow or when 128x128 tiles would leave most of the 256 CUs (2 blocks each) idle
    const bool small_n = N <= 64 || (long)fw_cdiv(M, 128) * fw_cdiv(N, 128) * splitk < 384;
    g_last_variant = (small_n ? 64 : 128) * 100 + (x_trans ? 10 : 0) + (w_trans ? 1 : 0);
    if (dtype == FW_DT_BF16) {
        return small_n ? dispatch_trans<bf16raw, 64>(a, x_trans, w_trans, st)
                       : dispatch_trans<bf16raw, 128>(a, x_trans, w_trans, st);
    }
    return small_n ? dispatch_trans<float, 64>(a, x_trans, w_trans, st)
                   : dispatch_trans<float, 128>(a, x_trans, w_trans, st);
}
