// Fused LeFF forward (K4 of SURVEY.md section 2.2; net/utils/leff.py:92-117):
//     y = x + rowscale * ( linear2( GELU( dwconv3x3( GELU( linear1(xn) ) ) ) ) )
// in ONE kernel per spatial tile, for the high-resolution stages (C <= 112, i.e. hidden 4C <= 448) where the unfused chain
// (GEMM + GELU twin -> depthwise 3x3 + GELU twin -> GEMM + residual) is bound by the six passes it makes over the [tokens][4C]
// hidden tensors.
//
// MI355X design.  A workgroup of 8 waves owns an 8 x 16 patch of pixels.
//   * The LayerNorm-ed input of the patch and its 1-pixel halo (10 x 18 pixels, C channels) is staged ONCE in LDS.
//   * The hidden dimension is walked in 7 chunks of CW = 4C/7 channels (16 / 32 / 64 for C = 28 / 56 / 112).  Per chunk:
//       GEMM 1 (MFMA): h1 = xn W1_chunk^T + b1 for all 192 (halo-padded) pixels; GELU; the result goes to LDS as the depthwise
//                      convolution's input (zero outside the image, as the reference's padding = 1 sees it);
//       depthwise 3x3 (VALU, operands from LDS): h2 = conv(g1) + bd; GELU; the result goes to LDS as GEMM 2's operand;
//       GEMM 2 (MFMA): y_acc += g2 W2[:, chunk]^T  -- the [128 pixels][C] output tile stays in registers across the 7 chunks.
//     The halo costs 1.5x on GEMM 1 and its GELU (192 instead of 128 pixels); nothing of the hidden tensor is ever re-read
//     from memory.
//   * Epilogue: + b2, DropPath row scale, + residual stream (f32).
//   * The backward pass still runs the unfused kernels, which need the pre- and post-activation tensors: h1, g1 (centre
//     pixels) are written by GEMM 1's epilogue and h2, g2 by the depthwise stage -- four streaming writes, no reads.  (A
//     recomputing backward would drop them: next step.)
// Weight fragments are read straight from L2 / L1 (W1: 4C x C, W2: C x 4C: 100 KB together at C = 112, shared by all workgroups).
#include "fw_common.h"
#include <stdlib.h>

namespace {

struct LeffArgs {
    const bf16raw* xn; long ldx;        // [T][C] LayerNorm output
    const bf16raw* w1;                  // [7 * CW][KP]  rows = hidden channel, K padded to KP = roundup(C, 32) with zeros
    const float* b1;                    // [4C]
    const float* wd;                    // [9][4C] depthwise taps, tap-major
    const float* bd;                    // [4C]
    const bf16raw* w2;                  // [CP][4C]  rows = output channel (CP = roundup(C, 16), zero rows), k = hidden
    const float* b2;                    // [C]
    const float* res; long ldr;         // f32 [T][C] residual stream
    const float* rowscale; int rows_per_scale;
    float* y; long ldy;                 // f32 [T][C]
    bf16raw* h1; bf16raw* g1; bf16raw* h2; bf16raw* g2; long ldh;   // [T][4C]
    int B, H, W, C;
};

constexpr int LT_Y = 8, LT_X = 16, LH_X = LT_X + 2, LH_N = (LT_Y + 2) * LH_X;     // 180 halo pixels, padded to 192 rows
constexpr int LTH = 512;

template <int CW>
__global__ __launch_bounds__(LTH, 2) void leff_fwd_kernel(LeffArgs a) {
    const bool twins = a.ldh > 0;                        // probe only (tools/leff_probe.py): ldh = 0 skips the four twin stores
    constexpr int CT = CW / 16;                          // MFMA column tiles per hidden chunk
    constexpr int KG = CW < 32 ? 32 : CW;                // k extent of GEMM 2's operand rows (zero-padded for CW = 16)
    constexpr int LDG1 = CW * 2 + 16;                    // g1 rows (bytes)
    constexpr int LDG2 = KG * 2 + 16;                    // g2 rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KP = (a.C + 31) & ~31, LDXS = KP * 2 + 16;
    char* xs = smem;                                     // [192][LDXS]
    char* g1s = xs + 192 * LDXS;                         // [192][LDG1]
    char* g2s = g1s + 192 * LDG1;                        // [128][LDG2]
    float* wds = reinterpret_cast<float*>(g2s + 128 * LDG2);     // [10][CW]: 9 taps + bias of the current chunk
    const int l = lane_id(), wv = threadIdx.x >> 6;
    const int C4 = 7 * CW;
    const int tx = a.W / LT_X, ty = a.H / LT_Y;
    const long ntile = (long)a.B * ty * tx;
    const int NT2 = (a.C + 15) / 16;                     // output column tiles (<= 7)
    // zero the k padding of g2s once (CW = 16: columns 16..31 stay zero for the whole kernel)
    if (KG != CW)
        for (int idx = threadIdx.x; idx < 128 * (KG - CW) / 8; idx += LTH)
            *reinterpret_cast<uint4*>(g2s + (idx / ((KG - CW) / 8)) * LDG2 + CW * 2 + (idx % ((KG - CW) / 8)) * 16) = make_uint4(0, 0, 0, 0);
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b = (int)(tile / (ty * tx)), r = (int)(tile % (ty * tx));
        const int y0 = (r / tx) * LT_Y, x0 = (r % tx) * LT_X;
        const long img0 = (long)b * a.H * a.W;
        __syncthreads();                                 // the previous tile's readers of xs are done
        // ---- stage xn of the halo patch (channels >= C and pixels outside the image / beyond 180: zero)
        {
            const int GR = KP / 8;
            for (int idx = threadIdx.x; idx < 192 * GR; idx += LTH) {
                const int p = idx / GR, g = idx % GR;
                const int iy = y0 - 1 + p / LH_X, ix = x0 - 1 + p % LH_X;
                const bool ok = p < LH_N && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && g * 8 < a.C;
                const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
                uint4 v = *reinterpret_cast<const uint4*>(a.xn + (img0 + (long)cy * a.W + cx) * a.ldx + (g * 8 < a.C ? g * 8 : 0));
                const int nv = a.C - g * 8;              // valid elements of this granule (>= 8: all)
                if (nv < 8) { if (nv <= 6) v.w = 0; if (nv <= 4) v.z = 0; if (nv <= 2) v.y = 0; if (nv & 1) { (nv == 1 ? v.x : nv == 3 ? v.y : nv == 5 ? v.z : v.w) &= 0xffffu; } }
                *reinterpret_cast<uint4*>(xs + p * LDXS + g * 16) = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
        f32x4 yacc[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) yacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float rs = a.rowscale ? a.rowscale[(img0) / a.rows_per_scale] : 1.0f;
#pragma unroll 1
        for (int j = 0; j < 7; ++j) {
            const int hc0 = j * CW;                      // first hidden channel of the chunk
            __syncthreads();                             // xs staged (j = 0) / the previous chunk's GEMM 2 has read g2s, its dwconv wds
            for (int idx = threadIdx.x; idx < 10 * CW; idx += LTH) {
                const int tap = idx / CW, c = idx % CW;
                wds[idx] = tap < 9 ? a.wd[(long)tap * C4 + hc0 + c] : a.bd[hc0 + c];
            }
            // ---- GEMM 1 + GELU: pairs (row tile of 16 halo pixels, column tile of 16 hidden channels) dealt to the 8 waves
            for (int pr = wv; pr < 12 * CT; pr += 8) {
                const int rt = pr / CT, ct = pr % CT;
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                const bf16raw* wrow = a.w1 + (long)(hc0 + ct * 16 + (l & 15)) * KP + ((l >> 4) << 3);
                for (int kc = 0; kc < KP / 32; ++kc) {
                    const uint4 af = *reinterpret_cast<const uint4*>(wrow + kc * 32);
                    const uint4 bf = frag_kc(xs, LDXS, rt * 16, kc);
                    mma_chunk<bf16raw>(acc, af, bf);
                }
                const int p = rt * 16 + (l & 15);                       // halo pixel of this lane's column
                const int py = p / LH_X, px = p % LH_X;
                const int iy = y0 - 1 + py, ix = x0 - 1 + px;
                const bool inimg = p < LH_N && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const int cl = ct * 16 + ((l >> 4) << 2);                // chunk-local hidden channel of acc[0]
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b1 + hc0 + cl);
                float hv[4], gv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { hv[q] = acc[q] + bb[q]; gv[q] = gelu_poly(hv[q]); }
                *reinterpret_cast<uint2*>(g1s + p * LDG1 + cl * 2) = inimg ? make_uint2(pack_bf2(gv[0], gv[1]), pack_bf2(gv[2], gv[3])) : make_uint2(0, 0);
                if (twins && inimg && py >= 1 && py <= LT_Y && px >= 1 && px <= LT_X) {      // centre pixel: this tile owns its h1 / g1
                    const long row = img0 + (long)iy * a.W + ix;
                    *reinterpret_cast<uint2*>(a.h1 + row * a.ldh + hc0 + cl) = make_uint2(pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3]));
                    *reinterpret_cast<uint2*>(a.g1 + row * a.ldh + hc0 + cl) = make_uint2(pack_bf2(gv[0], gv[1]), pack_bf2(gv[2], gv[3]));
                }
            }
            __syncthreads();
            // ---- depthwise 3x3 + GELU on the chunk: thread = (pixel, group of CW/4 channels)
            {
                constexpr int GCH = CW / 4;
                const int p = threadIdx.x >> 2, cg = (threadIdx.x & 3) * GCH;
                const int py = p / LT_X, px = p % LT_X;
                float s[GCH];
#pragma unroll
                for (int e = 0; e < GCH; ++e) s[e] = wds[9 * CW + cg + e];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const char* src = g1s + ((py + tap / 3) * LH_X + px + tap % 3) * LDG1 + cg * 2;
#pragma unroll
                    for (int e0 = 0; e0 < GCH; e0 += 4) {
                        const uint2 v = *reinterpret_cast<const uint2*>(src + e0 * 2);
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wds + tap * CW + cg + e0);
                        s[e0 + 0] += w4[0] * __uint_as_float(v.x << 16);
                        s[e0 + 1] += w4[1] * __uint_as_float(v.x & 0xffff0000u);
                        s[e0 + 2] += w4[2] * __uint_as_float(v.y << 16);
                        s[e0 + 3] += w4[3] * __uint_as_float(v.y & 0xffff0000u);
                    }
                }
                const long row = img0 + (long)(y0 + py) * a.W + x0 + px;
#pragma unroll
                for (int e0 = 0; e0 < GCH; e0 += 4) {
                    const float g0 = gelu_poly(s[e0]), g1v = gelu_poly(s[e0 + 1]), g2v = gelu_poly(s[e0 + 2]), g3 = gelu_poly(s[e0 + 3]);
                    const uint2 hp = make_uint2(pack_bf2(s[e0], s[e0 + 1]), pack_bf2(s[e0 + 2], s[e0 + 3]));
                    const uint2 gp = make_uint2(pack_bf2(g0, g1v), pack_bf2(g2v, g3));
                    *reinterpret_cast<uint2*>(g2s + p * LDG2 + (cg + e0) * 2) = gp;
                    if (twins) {
                        *reinterpret_cast<uint2*>(a.h2 + row * a.ldh + hc0 + cg + e0) = hp;
                        *reinterpret_cast<uint2*>(a.g2 + row * a.ldh + hc0 + cg + e0) = gp;
                    }
                }
            }
            __syncthreads();
            // ---- GEMM 2: wave w owns the 16 pixels of patch row w; y_acc[column tile] += W2[:, chunk] g2^T
            for (int kc = 0; kc < KG / 32; ++kc) {
                const uint4 bf = frag_kc(g2s, LDG2, wv * 16, kc);
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    if (t < NT2) {
                        const uint4 af = *reinterpret_cast<const uint4*>(a.w2 + (long)(t * 16 + (l & 15)) * C4 + hc0 + kc * 32 + ((l >> 4) << 3));
                        mma_chunk<bf16raw>(yacc[t], af, bf);
                    }
                }
            }
        }
        // ---- epilogue: y = res + rowscale * (acc + b2); lane holds 4 consecutive output channels of pixel (y0 + wv, x0 + (l & 15))
        const long row = img0 + (long)(y0 + wv) * a.W + x0 + (l & 15);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int c0 = t * 16 + ((l >> 4) << 2);
            if (t < NT2 && c0 < a.C) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b2 + c0);
                const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + row * a.ldr + c0);
                *reinterpret_cast<f32x4*>(a.y + row * a.ldy + c0) = rr + (yacc[t] + bb) * rs;
            }
        }
    }
}

template <int CW>
int leff_launch(const LeffArgs& a, hipStream_t st) {
    const int KP = (a.C + 31) & ~31, KG = CW < 32 ? 32 : CW;
    const size_t lds = (size_t)192 * (KP * 2 + 16) + 192 * (CW * 2 + 16) + 128 * (KG * 2 + 16) + 10 * CW * 4;
    FW_SET_LDS_ONCE((leff_fwd_kernel<CW>), 100 * 1024);
    FW_CHECK_ARG(lds <= 100 * 1024);
    const long ntile = (long)a.B * (a.H / LT_Y) * (a.W / LT_X);
    const int per_cu = 2 * lds <= 160 * 1024 ? 2 : 1;
    const int grid = (int)(ntile < 256 * per_cu ? ntile : 256 * per_cu);
    hipLaunchKernelGGL((leff_fwd_kernel<CW>), dim3(grid), dim3(LTH), lds, st, a);
    FW_LAUNCH_RET();
}

}  // namespace

// Fused LeFF forward, bf16 operands.  xn: [T][C] (ld ldx);  w1p: bf16 [4C][roundup(C, 32)] (zero-padded K);  wd: f32 [9][4C];
// w2p: bf16 [roundup(C, 16)][4C] (zero rows);  res / y: f32 [T][C];  h1, g1, h2, g2: bf16 [T][4C] (ld ldh) outputs kept for the
// backward pass.  C in {28, 56, 112} (hidden 4C = 7 chunks of 16 / 32 / 64), H % 8 == 0, W % 16 == 0.
extern "C" int fw_leff_fwd(const void* xn, long ldx, const void* w1p, const float* b1, const float* wd, const float* bd, const void* w2p,
                           const float* b2, const float* res, long ldr, const float* rowscale, int rows_per_scale, float* y, long ldy,
                           void* h1, void* g1, void* h2, void* g2, long ldh, int B, int H, int W, int C, void* stream) {
    FW_CHECK_ARG(xn && w1p && b1 && wd && bd && w2p && b2 && res && y && h1 && g1 && h2 && g2);
    FW_CHECK_ARG((C == 28 || C == 56 || C == 112) && H % 8 == 0 && W % 16 == 0 && B > 0);
    static const bool nostore = getenv("FW_LEFF_NOSTORE") != nullptr;
    if (nostore) ldh = 0;
    FW_CHECK_ARG(ldx % 8 == 0 && ldh % 4 == 0 && (ldh >= 4 * C || nostore) && ldr % 4 == 0 && ldy % 4 == 0 && (!rowscale || rows_per_scale > 0));
    FW_CHECK_ARG(((uintptr_t)xn & 15) == 0 && ((uintptr_t)w1p & 15) == 0 && ((uintptr_t)w2p & 15) == 0 && ((uintptr_t)res & 15) == 0 &&
                 ((uintptr_t)y & 15) == 0 && ((uintptr_t)b1 & 15) == 0 && ((uintptr_t)b2 & 15) == 0);
    LeffArgs a;
    a.xn = (const bf16raw*)xn; a.ldx = ldx; a.w1 = (const bf16raw*)w1p; a.b1 = b1; a.wd = wd; a.bd = bd; a.w2 = (const bf16raw*)w2p; a.b2 = b2;
    a.res = res; a.ldr = ldr; a.rowscale = rowscale; a.rows_per_scale = rows_per_scale; a.y = y; a.ldy = ldy;
    a.h1 = (bf16raw*)h1; a.g1 = (bf16raw*)g1; a.h2 = (bf16raw*)h2; a.g2 = (bf16raw*)g2; a.ldh = ldh; a.B = B; a.H = H; a.W = W; a.C = C;
    hipStream_t st = (hipStream_t)stream;
    if (C == 28) return leff_launch<16>(a, st);
    if (C == 56) return leff_launch<32>(a, st);
    return leff_launch<64>(a, st);
}
