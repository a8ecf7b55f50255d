// 3x3 convolution forward / data gradient in the Winograd F(4,3) domain on the bf16 MFMA with EXACT three-way bf16
// operand splits (fp32-class results) for gfx950.  Same op sites and contract as conv_wino.hip (pb_sed/models/weak_label/
// crnn.py:93; the MFMA-bound 3x3 layers of pb_sed/experiments/weak_label_crnn/training.py:159-169), same tensors, same
// prologue / epilogue fusions.  What changes is where the multiplications run: conv_wino.hip issues them on the fp32 MFMA,
// which shares the SIMD's pipe with the VALU and runs at 1/16 of the bf16 rate; here every transformed operand is split
// exactly into three bf16 parts (x = hi + mid + lo by truncation, 8 + 8 + 8 significant bits) and the six part products above
// 2^-24 are accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (Bf3 / mfma_x3 in common.h): 6/16 of the fp32 pipe time, and
// the staging VALU no longer competes with the MFMAs.
//
//   M_xi[cout, (row f, tile i)] = sum_{kh, cin} U_xi[kh][cout, cin] * V_xi[cin, (row f + kh - 1, tile i)],   y = A^T M
//
// Block = 512 threads = 4 CONSUMER waves + 4 PRODUCER waves (one of each per SIMD), tile = 64 cout x 4 rows x 64 t:
//  * consumer wave w owns 16 cout x 4 rows x 16 tiles and all 6 transform points (96 accumulator registers), so the output
//    transform is register arithmetic as in conv_wino.hip.  Its A operands - the transformed, pre-split weights U - are NOT
//    staged: the pack writes them in fragment order ([chunk][xi][kh][cout tile][part][lane] x 16 B) and the wave streams its
//    own 1 KB pieces from L2 straight into a register ring of three transform points (loads issued two points = 144 MFMAs
//    ahead; one point ahead the wave drains its whole request queue at every point boundary: 67 % of the MFMA peak).
//    U is 108 B per weight pair; through LDS it would have cost 221 KB of writes per 32-channel chunk for two reads each.
//  * producer waves turn the raw input into V: global load (BN-apply + ReLU + mask prologue, un-pool for pooled data
//    gradients) -> B^T d -> three bf16 parts -> LDS image [xi][halo row][tile][32 cin] per part (64-byte positions,
//    XOR-swizzled 16-byte groups: a B fragment is one conflict-free ds_read_b128 per part).  The image of a 32-channel
//    chunk is built in two halves (xi 0..2, xi 3..5; 54 KB each, double-buffered): while the consumers multiply one half
//    the producers build the other, one block barrier per half.  What the split buys is a bubble-free pipeline (no wave
//    ever waits for its own staging), NOT free VALU: on gfx950 the VALU instructions of one wave and the bf16 MFMAs of
//    another wave on the same SIMD take turns (tools/micro/mfma_valu_bf16.hip: 72 MFMAs 1 240 clocks, 256 v_fma of the
//    partner wave 681, together 1 769 = 0.92 x the sum), so every producer instruction is paid at ~2.5 clocks of MFMA
//    time - the producer is written for instruction count: one pass over a chunk's registers for all six points, the
//    second half's packed words parked in registers, 8-byte LDS stores, the row's outer elements in two loads.
// Blocks are PERSISTENT (one per CU) and walk a list of (cout tile, spatial tile) items; the producers run ahead through
// the chunk stream of all of a block's tiles, so a tile's first half image is built during the previous tile's last phase
// and epilogue.  Items are numbered so that an XCD works on one cout tile (item % 8 = XCD): its L2 holds that tile's U only.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "pack_elems.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int WX_CT = 64;                               // cout per block (16 per consumer wave)
constexpr int WX_CK = 32;                               // cin per chunk = K of one MFMA
constexpr int WX_FT = 4, WX_TT = 64, WX_ROWS = WX_FT + 2;
constexpr int WX_PART = 3 * WX_ROWS * 16 * 64;          // bytes of one part of a half image: [xi 3][row 6][tile 16][32 cin bf16]
constexpr int WX_HALF = 3 * WX_PART;                    // 55 296
constexpr int WX_V_BYTES = 2 * WX_HALF;
#ifndef WX_DBG
#define WX_DBG 0            // ablation switches of tools/kernel_ablation.sh (never set in the product build)
#endif
#ifndef WX_RING_N
#define WX_RING_N 2
#endif
constexpr int WX_RING = WX_RING_N;                      // U register ring in transform points (3 kh x 3 parts x 16 B per lane each)

template <bool POOL>
struct WxCfg {
    static constexpr int FO_T = POOL ? WX_FT / 2 : WX_FT;
    static constexpr size_t LDS_BYTES = (size_t)WX_V_BYTES;
};

__global__ void winox3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ up, int Cout, int Cin, int InP,
                                   int OutP, int dgrad) {
    const size_t total = (size_t)18 * InP * OutP;               // weights; three halfwords each
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        pack_winox3_value(w, up, i, Cout, Cin, InP, OutP, dgrad);
}

// CT: cout per block.  64: consumer wave w owns cout tile w (16 cout) and all 4 rows.  32 (launches that produce <= 32
// channels - with 64-cout blocks two of the four consumer waves would multiply zero padding): wave w owns cout tile w & 1 and
// the row pair w >> 1, i.e. 16 cout x 2 rows x 16 tiles x 6 points (48 accumulator registers) and 4 of the 6 halo rows.
template <bool POOL, bool DGRAD, bool UNPOOL, int CT = WX_CT>
__global__ __launch_bounds__(512) void conv_winox3_kernel(ConvFwdArgs a, int nCt, int nSp, int xcd_map, int nWork) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // uniform: buffer-load scalar offsets and role branches depend on it
    const bool consumer = wave < 4;
    const int lq = lane >> 4, lr = lane & 15;

    // PERSISTENT blocks: block p works on items p, p + gridDim, p + 2 gridDim .. of the (cout tile, spatial tile) list.
    // xcd_map: item -> XCD = item % 8 (gridDim is a multiple of 8), and an XCD keeps ONE cout tile: its L2 holds that tile's U.
    const int nTt = (a.T + WX_TT - 1) / WX_TT, nFt = (a.F + WX_FT - 1) / WX_FT;
    auto item = [&](int k, int& ct, int& sp) __attribute__((always_inline)) {
        const int w = (int)blockIdx.x + k * (int)gridDim.x;
        if (xcd_map == 1) {                        // an XCD keeps one cout tile (its L2 holds that tile's U only)
            const int xcd = w & 7, slot = w >> 3, per = 8 / nCt;
            ct = xcd % nCt;
            sp = slot * per + xcd / nCt;
        } else if (xcd_map == 2) {                 // the cout tiles of a spatial tile run side by side on ONE XCD: x is fetched from
            const int xcd = w & 7, l = w >> 3;     // HBM once and found in that XCD's L2 by the others (and so are the halo rows of
            ct = l % nCt;                          // the row tile above / below when the row has 8 column tiles)
            sp = (l / nCt) * 8 + xcd;
        } else if (xcd_map == 3) {                 // an XCD takes whole CLIPS (b % 8 = XCD) and all cout tiles of a spatial tile side
            const int xcd = w & 7, l = w >> 3;     // by side: the cache lines neighbouring column tiles share, the halo rows of the row
            ct = l % nCt;                          // tiles above / below and x for the other cout tiles are all found in that XCD's L2
            const int spl = l / nCt, per = nTt * nFt;
            const int b = (spl / per) * 8 + xcd;
            sp = b < a.B ? b * per + spl % per : nSp;
        } else {
            ct = w % nCt;
            sp = w / nCt;
        }
        return w < nWork && sp < nSp;
    };
    int nT = 0;                                       // valid items of this block (sp grows with k: the invalid ones are at the end)
    {
        int ct_, sp_;
        while (item(nT, ct_, sp_)) ++nT;
    }
    if (nT == 0) return;
    const bool pro = a.scale != nullptr;
    constexpr bool unpool = DGRAD && UNPOOL;
    const int Fsrc = unpool ? a.F / 2 : a.F;
    const int nChunks = a.CinP / WX_CK;
    const int G = nT * nChunks;                       // the block's stream of 32-channel chunks over all of its tiles

    if (!consumer) {
        // ================================================================ PRODUCER: x -> V (three bf16 parts) in LDS
        // Item = (4 consecutive cin, halo row, tile): a tile's four inputs are ONE aligned 16-byte load per channel (16 lanes =
        // 16 tiles = 256 contiguous bytes of a row), the two inputs it shares with its neighbours (d0 = x[4i-1], d5 = x[4i+4])
        // come from the neighbouring lanes by DPP after the prologue; only the row's first / last tile loads its outer
        // element itself.  Four channels per item make every LDS store an 8-byte one (4 cin x bf16 of one transform point and
        // part): 27 stores per thread and half image instead of 108 two-byte ones.
        // A thread's 3 items: cin group lane / 16 + 4 (producer wave & 1) (4 channels), rows 2 j + (producer wave >> 1).
        // The producers run through the chunk stream of ALL tiles of the block: the first half image of the next tile is
        // built while the consumers are still in the last phase / the epilogue of the current one.
        const int pw = wave - 4, tile = lr;
        // (lanes 0..31 of a wave = the two 4-channel halves of ONE 16-byte k-group: their 8-byte stores cover all 64 banks)
#if WX_DBG & 32                                                      // ablation: the mapping whose stores are two-way bank conflicts
        const int cgp = (pw * 4 + lq) >> 1, rsel = (pw * 4 + lq) & 1;
#else
        const int cgp = lq | ((pw & 1) << 2), rsel = pw >> 1;
#endif
        constexpr unsigned OOB = 0x80000000u;
        const unsigned clip_elems = (unsigned)(a.Cin * Fsrc * a.T);
        const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.scale), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.shift), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        const bool edge_lane = tile == 0 || tile == 15;
        const float relu_floor = (pro && a.relu) ? 0.f : -__builtin_inff();
        const unsigned row_elems = (unsigned)a.T;
        const unsigned chan_step = (unsigned)(Fsrc * a.T);
        // LDS byte offset of this thread's 8-byte slot in a (point, row) plane: tile position, swizzled k-group, half
        const unsigned lds_t = (unsigned)(tile * 64 + ((((cgp >> 1) ^ ((-(tile >> 2)) & 3)) & 3) * 16) + (cgp & 1) * 8 + rsel * 1024);

        // state of the tile the NEXT load_chunk reads from (set when the stream enters a tile) ...
        int ld_k = 0, ld_ch = 0, ld_f0 = 0, ld_t0 = 0, ld_b = 0;
        // ... and of the tile a raw buffer's registers belong to.  NB raw buffers = chunks whose loads are in flight (two; the
        // un-pooling variant, which also carries index bytes, one).  A chunk's registers are consumed in ONE pass (`produce`):
        // prologue, DPP neighbours, all six transform points, splits - the first half image goes to LDS at once, the packed
        // 8-byte words of the second half wait in 54 registers and are only STORED a phase later.  (Rebuilding the inputs for
        // each half cost 40 % more VALU instructions, and the producers' VALU instructions share each SIMD's issue port with
        // the consumer wave's MFMAs: PMC - MFMA pipe 47 % busy, 1 450 VALU instructions per producer wave and chunk.)
        constexpr int NB = UNPOOL ? 1 : 2;
        int fi_f0[NB], fi_nok[NB];
        bool fi_edge_ok[NB];
        uint2 keep[3][3][3];                                    // [item][point of the second half][part]

        unsigned rin[NB][3][4][4], redgeL[NB], redgeR[NB], ridx[NB][3][4], ridx_eL[NB], ridx_eR[NB];
        u32x4_t rsc[NB], rsh[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) { rsc[n] = u32x4_t{0u, 0u, 0u, 0u}; rsh[n] = u32x4_t{0u, 0u, 0u, 0u}; fi_f0[n] = 0; fi_nok[n] = 0; fi_edge_ok[n] = false; }

        auto load_chunk = [&](auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            if (ld_ch == 0) {                                      // entering tile ld_k
                int ct, sp;
                item(ld_k, ct, sp);
                ld_t0 = (sp % nTt) * WX_TT;
                ld_f0 = ((sp / nTt) % nFt) * WX_FT;
                ld_b = sp / (nTt * nFt);
            }
            const int f0 = ld_f0, t0 = ld_t0;
            const int sl = a.seq_len ? min(a.seq_len[ld_b], a.T) : a.T;
            const int tlim = pro ? sl : a.T;                       // Normalization re-masks its output (y*mask)
            const int tq = t0 + 4 * tile;                          // first output column of the tile = its input d1
            const int t_edge = tile == 0 ? t0 - 1 : t0 + 64;      // the row's outer element tiles 0 / 15 need
            fi_f0[BUF] = f0;
            fi_nok[BUF] = min(max(tlim - tq, 0), 4);               // d1..d4 inside [0, tlim)  (zero padding is post-activation)
            fi_edge_ok[BUF] = edge_lane && t_edge >= 0 && t_edge < tlim;
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)ld_b * clip_elems, 0, clip_elems * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
                unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)ld_b * clip_elems : nullptr, 0, unpool ? clip_elems : 0u, 0x00020000);
            const unsigned cin0 = (unsigned)(ld_ch * WX_CK + 4 * cgp);
            {                                                      // four consecutive channels (Cin % 4 == 0 is not required: element
                rsc[BUF].x = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, cin0 * 4u, 0, 0);          // loads, range-checked; no prologue: empty)
                rsc[BUF].y = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, cin0 * 4u + 4u, 0, 0);
                rsc[BUF].z = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, cin0 * 4u + 8u, 0, 0);
                rsc[BUF].w = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, cin0 * 4u + 12u, 0, 0);
                rsh[BUF].x = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, cin0 * 4u, 0, 0);
                rsh[BUF].y = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, cin0 * 4u + 4u, 0, 0);
                rsh[BUF].z = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, cin0 * 4u + 8u, 0, 0);
                rsh[BUF].w = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, cin0 * 4u + 12u, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int fin = f0 - 1 + 2 * j + rsel;
                const bool row_ok = fin >= 0 && fin < a.F;
                const unsigned row0 = cin0 * chan_step + (unsigned)(unpool ? (fin >> 1) : fin) * row_elems;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned e1 = row_ok ? row0 + (unsigned)c * chan_step + (unsigned)tq : (OOB >> 2);           // element index of d1
                    const u32x4_t xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, e1 * 4u, 0, 0);
                    rin[BUF][j][c][0] = xv.x; rin[BUF][j][c][1] = xv.y; rin[BUF][j][c][2] = xv.z; rin[BUF][j][c][3] = xv.w;
                    if (unpool) ridx[BUF][j][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, e1 >= (OOB >> 2) ? OOB : e1, 0, 0);
                }
            }
            // The outer elements of the row (x[t0 - 1] for tile 0, x[t0 + 64] for tile 15) of all 12 (row j, channel c) pairs of
            // this 16-lane group in TWO loads: lane k < 12 fetches the left one of pair k = 4 j + c, lane k + 4 the right one;
            // `produce` moves them to lanes 0 / 15 with DPP row shifts.  (One load per pair and side - 24 instructions with two
            // active lanes each - kept the CU's texture path as busy as all the 16-byte loads together.)
            {
                const int kl = tile, kr = tile - 4;
                const int jl = kl >> 2, cl_ = kl & 3, jr = kr >> 2, cr = kr & 3;
                const int finl = f0 - 1 + 2 * jl + rsel, finr = f0 - 1 + 2 * jr + rsel;
                const bool okl = kl < 12 && finl >= 0 && finl < a.F && t0 - 1 >= 0;
                const bool okr = kr >= 0 && finr >= 0 && finr < a.F && t0 + 64 < a.T;
                const unsigned el = okl ? (cin0 + (unsigned)cl_) * chan_step + (unsigned)(unpool ? (finl >> 1) : finl) * row_elems + (unsigned)(t0 - 1) : (OOB >> 2);
                const unsigned er = okr ? (cin0 + (unsigned)cr) * chan_step + (unsigned)(unpool ? (finr >> 1) : finr) * row_elems + (unsigned)(t0 + 64) : (OOB >> 2);
                redgeL[BUF] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, el * 4u, 0, 0);
                redgeR[BUF] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, er * 4u, 0, 0);
                if (unpool) {
                    ridx_eL[BUF] = __builtin_amdgcn_raw_buffer_load_b8(rs_i, el >= (OOB >> 2) ? OOB : el, 0, 0);
                    ridx_eR[BUF] = __builtin_amdgcn_raw_buffer_load_b8(rs_i, er >= (OOB >> 2) ? OOB : er, 0, 0);
                }
            }
            if (!(WX_DBG & 16) && ++ld_ch == nChunks) { ld_ch = 0; ++ld_k; }     // ablation bit 16: every chunk re-reads the first one
        };
        // raw registers of buffer BUF -> post-activation inputs -> V of all six transform points: points 0..2 -> LDS half image 0,
        // points 3..5 -> `keep`
        auto produce = [&](auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            // outer elements of all 12 (j, c) pairs: pair k's left one sits in lane k, its right one in lane k + 4 (load_chunk)
            float edge[12];
            unsigned edge_i[12];
            {
                const int eL = (int)redgeL[BUF], eR = (int)redgeR[BUF], iL = (int)ridx_eL[BUF], iR = (int)ridx_eR[BUF];
#define WX_EDGE(K)                                                                                                               \
                {                                                                                                                \
                    const int l = K == 0 ? eL : __builtin_amdgcn_update_dpp(0, eL, 0x100 + (K == 0 ? 1 : K), 0xf, 0xf, true);    /* row_shl:K -> lane 0 */ \
                    const int r = K == 11 ? eR : __builtin_amdgcn_update_dpp(0, eR, 0x110 + (K == 11 ? 1 : 11 - K), 0xf, 0xf, true); /* row_shr:11-K -> lane 15 */ \
                    edge[K] = __int_as_float(tile == 0 ? l : r);                                                                 \
                    if (unpool) {                                                                                                \
                        const int li = K == 0 ? iL : __builtin_amdgcn_update_dpp(0, iL, 0x100 + (K == 0 ? 1 : K), 0xf, 0xf, true); \
                        const int ri = K == 11 ? iR : __builtin_amdgcn_update_dpp(0, iR, 0x110 + (K == 11 ? 1 : 11 - K), 0xf, 0xf, true); \
                        edge_i[K] = (unsigned)(tile == 0 ? li : ri);                                                             \
                    }                                                                                                            \
                }
                WX_EDGE(0) WX_EDGE(1) WX_EDGE(2) WX_EDGE(3) WX_EDGE(4) WX_EDGE(5) WX_EDGE(6) WX_EDGE(7) WX_EDGE(8) WX_EDGE(9) WX_EDGE(10) WX_EDGE(11)
#undef WX_EDGE
            }
            // prologue without selects: x * 1 + 0 when there is none (the empty scale / shift resources read 0), ReLU as a max with
            // 0 or -inf
            const float scv[4] = {pro ? __uint_as_float(rsc[BUF].x) : 1.f, pro ? __uint_as_float(rsc[BUF].y) : 1.f,
                                  pro ? __uint_as_float(rsc[BUF].z) : 1.f, pro ? __uint_as_float(rsc[BUF].w) : 1.f};
            const float shv[4] = {__uint_as_float(rsh[BUF].x), __uint_as_float(rsh[BUF].y), __uint_as_float(rsh[BUF].z), __uint_as_float(rsh[BUF].w)};
            unsigned char* base = smem_raw + lds_t;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int fin = fi_f0[BUF] - 1 + 2 * j + rsel;
                const int par = fin & 1;
                const bool live = fin >= 0 && fin < a.F;          // a halo row outside the plane is zero AFTER the activation
                float d[4][6];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float sc = scv[c], sh = shv[c];
                    float v[4], ve = edge[j * 4 + c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float u = __uint_as_float(rin[BUF][j][c][k]);
                        if (unpool) u = (int)((ridx[BUF][j][c] >> (8 * k)) & 0xffu) != par ? 0.f : u;
                        u = fmaxf(fmaf(u, sc, sh), relu_floor);
                        v[k] = (live && k < fi_nok[BUF]) ? u : 0.f;
                    }
                    if (unpool) ve = (int)(edge_i[j * 4 + c] & 0xffu) != par ? 0.f : ve;
                    ve = fmaxf(fmaf(ve, sc, sh), relu_floor);
                    ve = (live && fi_edge_ok[BUF]) ? ve : 0.f;
                    // d0 = the left neighbour's last input, d5 = the right neighbour's first one
                    const float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[3]), 0x111, 0xf, 0xf, true));    // row_shr:1
                    const float right = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0x101, 0xf, 0xf, true));   // row_shl:1
                    d[c][0] = tile == 0 ? ve : left;
                    d[c][1] = v[0]; d[c][2] = v[1]; d[c][3] = v[2]; d[c][4] = v[3];
                    d[c][5] = tile == 15 ? ve : right;
                }
                unsigned char* pj = base + j * 2048;
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    unsigned hi[4], mid[4], lo[4];                      // bit patterns; the bf16 part is the upper half
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float d0 = d[c][0], d1 = d[c][1], d2 = d[c][2], d3 = d[c][3], d4 = d[c][4], d5 = d[c][5];
                        const float v = x == 0 ? 4.f * d0 - 5.f * d2 + d4
                                      : x == 1 ? -4.f * (d1 + d2) + d3 + d4
                                      : x == 2 ? 4.f * (d1 - d2) - d3 + d4
                                      : x == 3 ? -2.f * d1 - d2 + 2.f * d3 + d4
                                      : x == 4 ? 2.f * d1 - d2 - 2.f * d3 + d4
                                               : 4.f * d1 - 5.f * d3 + d5;
                        const unsigned u0 = __float_as_uint(v);
                        const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
                        const unsigned u1 = __float_as_uint(r1);
                        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                        hi[c] = u0; mid[c] = u1; lo[c] = __float_as_uint(r2);
                    }
                    // upper halves of (c0, c1) and (c2, c3) -> two dwords, channel c0 in the low half (v_perm_b32)
                    const uint2 w0 = make_uint2(__builtin_amdgcn_perm(hi[1], hi[0], 0x07060302u), __builtin_amdgcn_perm(hi[3], hi[2], 0x07060302u));
                    const uint2 w1 = make_uint2(__builtin_amdgcn_perm(mid[1], mid[0], 0x07060302u), __builtin_amdgcn_perm(mid[3], mid[2], 0x07060302u));
                    const uint2 w2 = make_uint2(__builtin_amdgcn_perm(lo[1], lo[0], 0x07060302u), __builtin_amdgcn_perm(lo[3], lo[2], 0x07060302u));
                    if (x < 3) {
                        unsigned char* p = pj + x * (WX_ROWS * 16 * 64);
                        *reinterpret_cast<uint2*>(p) = w0;
                        *reinterpret_cast<uint2*>(p + WX_PART) = w1;
                        *reinterpret_cast<uint2*>(p + 2 * WX_PART) = w2;
                    } else {
                        keep[j][x - 3][0] = w0; keep[j][x - 3][1] = w1; keep[j][x - 3][2] = w2;
                    }
                }
            }
        };
        auto flush_half1 = [&]() __attribute__((always_inline)) {      // the kept words of points 3..5 -> LDS half image 1
            unsigned char* base = smem_raw + WX_HALF + lds_t;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int xl = 0; xl < 3; ++xl) {
                    unsigned char* p = base + j * 2048 + xl * (WX_ROWS * 16 * 64);
                    *reinterpret_cast<uint2*>(p) = keep[j][xl][0];
                    *reinterpret_cast<uint2*>(p + WX_PART) = keep[j][xl][1];
                    *reinterpret_cast<uint2*>(p + 2 * WX_PART) = keep[j][xl][2];
                }
        };

        constexpr bool P_LD = !(WX_DBG & 1), P_ST = !(WX_DBG & 2);
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, NB - 1>;
        // chunk c of the stream lives in raw buffer c % NB and is re-loaded with chunk c + NB as soon as it is consumed
        if (P_LD) load_chunk(I0{});
        if (NB > 1 && G > 1 && P_LD) load_chunk(I1{});
        if (P_ST) produce(I0{});
        if (G > NB && P_LD) load_chunk(I0{});
        __syncthreads();
        auto step = [&](int g, auto nxt_c) __attribute__((always_inline)) {     // NXT: raw buffer of chunk g + 1
            using NXT = decltype(nxt_c);
            if (P_ST) flush_half1();                                    // consumers: xi 0..2 of chunk g
            __syncthreads();
            if (g + 1 < G) {                                            // consumers: xi 3..5 of chunk g (then, at a tile's end, its epilogue)
                if (P_ST) produce(NXT{});
                if (g + 1 + NB < G && P_LD) load_chunk(NXT{});
            }
            __syncthreads();
        };
        if (NB == 1) {
            for (int g = 0; g < G; ++g) step(g, I0{});
        } else {
            for (int g = 0; g < G; g += 2) {
                step(g, I1{});
                if (g + 1 < G) step(g + 1, I0{});
            }
        }
        return;
    }

    // ================================================================ CONSUMER: U from L2, V from LDS, MFMAs, epilogue
    const int MT = a.CoutP / 16;
    const size_t u_bytes = (size_t)18 * a.CinP * a.CoutP * 3 * 2;
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (unsigned)u_bytes, 0x00020000);
    const unsigned voff_u = (unsigned)lane * 16u;
    const unsigned step_bytes = (unsigned)MT * 3072u;               // one (xi, kh) step of all cout tiles
    const unsigned point_bytes = 3u * step_bytes;
    const unsigned v_lane = (unsigned)(lr * 64 + (((lq ^ ((-(lr >> 2)) & 3)) & 3) * 16));

    u32x4_t A[WX_RING][3][3];                                        // [ring slot][kh][part]
    auto load_A = [&](int slot, unsigned soff) __attribute__((always_inline)) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                A[slot][kh][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u, soff + kh * step_bytes + p * 1024u, 0);
    };
    constexpr bool H32 = CT == 32;
    constexpr int NF = H32 ? 2 : WX_FT;                              // output rows of a consumer wave
    const int cw = H32 ? (wave & 1) : wave;                          // its 16-cout tile inside the block
    const int f_lo = H32 ? (wave >> 1) * 2 : 0;                      // its first output row
    int ct, sp;
    item(0, ct, sp);
    unsigned u_base = (unsigned)(ct * (CT / 16) + cw) * 3072u;       // point 0 of chunk 0, this wave's cout tile
    static_assert(WX_RING == 2 || WX_RING == 3, "the ring holds the current point and the next one (two)");
    load_A(0, u_base);
    if (WX_RING == 3) load_A(1, u_base + point_bytes);

    f32x4 acc[6][NF];
#if WX_DBG & 128
    // profiling build: shader-clock time of consumer wave 0 of block 0 in (a) the MFMA phases, (b) the block barriers, (c) the
    // epilogues; written over the first floats of y at the end (thousands of clocks)
    unsigned long long pt_mfma = 0, pt_bar = 0, pt_epi = 0, pt_t = __builtin_amdgcn_s_memtime();
    const unsigned long long pt_begin = pt_t;
#define WX_TICK(acc_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - pt_t; pt_t = now_; }
#else
#define WX_TICK(acc_)
#endif
    __syncthreads();                                                 // half 0 of the first chunk is staged
    WX_TICK(pt_bar)
    for (int k = 0; k < nT; ++k) {
        const int t0 = (sp % nTt) * WX_TT;
        const int f0 = ((sp / nTt) % nFt) * WX_FT;
        const int b = sp / (nTt * nFt);
        const int cout0 = ct * CT;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        int ct_n = ct, sp_n = sp;
        const bool more = k + 1 < nT && item(k + 1, ct_n, sp_n);
        const unsigned u_base_n = more ? (unsigned)(ct_n * (CT / 16) + cw) * 3072u : 0xC0000000u;       // past the end: reads 0
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int fl = 0; fl < NF; ++fl) acc[x][fl] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nChunks; ++ch) {
            const unsigned soff_u = u_base + (unsigned)ch * 6u * point_bytes;
            const unsigned soff_next = ch + 1 < nChunks ? soff_u + 6u * point_bytes : u_base_n;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned char* vb = smem_raw + half * WX_HALF + v_lane + f_lo * 1024;     // the wave's first halo row
                u32x4_t Bf[2][3];                                        // V fragments of two halo rows: the next row's reads are
                auto read_B = [&](int xl, int h, u32x4_t (&dst)[3]) __attribute__((always_inline)) {   // in flight during a row's MFMAs
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        dst[p] = *reinterpret_cast<const u32x4_t*>(vb + p * WX_PART + (xl * WX_ROWS + h) * 1024);
                };
                read_B(0, 0, Bf[0]);
#pragma unroll
                for (int xl = 0; xl < 3; ++xl) {
                    const int x = half * 3 + xl;
                    // U of the next point into the slot the previous point left
                    constexpr int AHEAD = WX_RING - 1;                  // points between a load and its use
                    if (!(WX_DBG & 4))
                        load_A((x + AHEAD) % WX_RING, x + AHEAD < 6 ? soff_u + (unsigned)(x + AHEAD) * point_bytes
                                                                    : soff_next + (unsigned)(x + AHEAD - 6) * point_bytes);
                    if (WX_DBG & 8) continue;
#pragma unroll
                    for (int h = 0; h < NF + 2; ++h) {                     // halo rows f_lo + h of the wave's NF output rows
                        const int cur = (xl * (NF + 2) + h) & 1;
                        if (h + 1 < NF + 2) read_B(xl, h + 1, Bf[cur ^ 1]);
                        else if (xl + 1 < 3) read_B(xl + 1, 0, Bf[cur ^ 1]);
                        // six part products, smallest first, round-robin over the (row, kh) sets of this halo row
#pragma unroll
                        for (int pp = 0; pp < 6; ++pp) {
                            const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;     // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                            const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh) {
                                const int fl = h - kh;
                                if (fl >= 0 && fl < NF) acc[x][fl] = mfma_b16(A[x % WX_RING][kh][pa], Bf[cur][pb], acc[x][fl]);
                            }
                        }
                    }
                }
                WX_TICK(pt_mfma)
                __syncthreads();
                WX_TICK(pt_bar)
            }
        }

        // ---- epilogue of this tile (the producers are already staging the next one): A^T M in registers, then bias / pool /
        // statistics / BN-ReLU backward as in conv_wino.hip on 4 consecutive t per lane:
        //   t = t0 + 4*lr + e,  cout = cout0 + 16*wave + lq*4 + r,  f = f0 + fl.
        // A wave owns its 16 channels alone, so the statistics need no cross-wave pass: one lane per (channel[, row]) adds them.
        constexpr unsigned OOB_C = 0x80000000u, OOB_T = 0x20000000u;
        constexpr int FO_T = POOL ? WX_FT / 2 : WX_FT;
        const int Fo = POOL ? a.F / 2 : a.F;
        const int tb = t0 + 4 * lr;
        const unsigned oclip = (unsigned)(a.Cout * Fo * a.T);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * oclip, 0, oclip * 4u, 0x00020000);
        const bool want_idx = POOL && a.pool_idx != nullptr;
        const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
            want_idx ? a.pool_idx + (size_t)b * oclip : nullptr, 0, want_idx ? oclip : 0u, 0x00020000);
        const bool bnb = DGRAD && a.bx != nullptr;
        const __amdgpu_buffer_rsrc_t rs_bx = __builtin_amdgcn_make_buffer_rsrc(
            bnb ? const_cast<float*>(a.bx) + (size_t)b * oclip : nullptr, 0, bnb ? oclip * 4u : 0u, 0x00020000);
        const unsigned tcol = tb < a.T ? (unsigned)tb * 4u : OOB_T;
        const int n_seq = min(max(sl - tb, 0), 4);                    // elements of the quad inside the sequence
        const int slot = (int)(sp & (PBSED_STAT_SLOTS - 1));
        constexpr int NFO = POOL ? 1 : 2;                             // output rows per row pair
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = cout0 + cw * 16 + lq * 4 + r;
            const bool cv = cout < a.Cout;
            const float bias = (a.bias && cv) ? a.bias[cout] : 0.f;
            const unsigned coff = cv ? (unsigned)(cout * Fo * a.T) * 4u : OOB_C;
            float bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
            if (bnb && cv) { bsc = a.bscale[cout]; bsh = a.bshift[cout]; bmu = a.bmean[cout]; bis = a.binvstd[cout]; }
            float c1 = 0.f, c2 = 0.f;                                   // per-channel statistics over the block's rows
#pragma unroll
            for (int rpl = 0; rpl < NF / 2; ++rpl) {                    // the wave's row pairs ((0,1), (2,3)): one pool window each
                const int rp = rpl + f_lo / 2;
                unsigned roff[NFO];
                bool orow_ok[NFO];
                int fo_of[NFO];
#pragma unroll
                for (int fo_l = 0; fo_l < NFO; ++fo_l) {
                    const int fo = POOL ? f0 / 2 + rp : f0 + rp * 2 + fo_l;
                    fo_of[fo_l] = fo;
                    orow_ok[fo_l] = fo < Fo;
                    roff[fo_l] = orow_ok[fo_l] ? (unsigned)(fo * a.T) * 4u : OOB_T;
                }
                u32x4_t xq[NFO];
                if (bnb) {
#pragma unroll
                    for (int fo_l = 0; fo_l < NFO; ++fo_l) xq[fo_l] = __builtin_amdgcn_raw_buffer_load_b128(rs_bx, coff + roff[fo_l] + tcol, 0, 0);
                }
                float y[2][4];
#pragma unroll
                for (int fl = 0; fl < 2; ++fl) {
                    const int f = rpl * 2 + fl;
                    const float m0 = acc[0][f][r], m1 = acc[1][f][r], m2 = acc[2][f][r], m3 = acc[3][f][r], m4 = acc[4][f][r],
                                m5 = acc[5][f][r];
                    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                    y[fl][0] = m0 + s12 + s34 + bias;
                    y[fl][1] = d12 + 2.f * d34 + bias;
                    y[fl][2] = s12 + 4.f * s34 + bias;
                    y[fl][3] = d12 + 8.f * d34 + m5 + bias;
                }
#pragma unroll
                for (int fo_l = 0; fo_l < NFO; ++fo_l) {
                    float v[4];
                    unsigned pbytes = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (POOL) {
                            const bool second = y[1][e] > y[0][e];
                            v[e] = second ? y[1][e] : y[0][e];
                            pbytes |= (unsigned)second << (8 * e);
                        } else {
                            v[e] = y[fo_l][e];
                        }
                    }
                    const int n_cnt = orow_ok[fo_l] ? n_seq : 0;           // padded channels produce exact zeros
                    float s1 = 0.f, s2 = 0.f;
                    if (DGRAD) {
                        if (bnb) {
                            // backward through mask -> ReLU -> BN-apply of the layer's prologue, with the BN-backward sums
                            const u32x4_t x4 = xq[fo_l];
                            const float xv[4] = {__uint_as_float(x4.x), __uint_as_float(x4.y), __uint_as_float(x4.z), __uint_as_float(x4.w)};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float z = fmaf(xv[e], bsc, bsh);
                                const bool keep = e < n_cnt && (!a.relu || z > 0.f);
                                v[e] = keep ? v[e] : 0.f;
                                s1 += v[e]; s2 = fmaf(v[e], (xv[e] - bmu) * bis, s2);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float vm = e < n_cnt ? v[e] : 0.f;
                            s1 += vm; s2 = fmaf(vm, vm, s2);
                        }
                    }
                    const unsigned off = coff + roff[fo_l] + tcol;
                    const u32x4_t q = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(q, rs_y, off, 0, 0);
                    if (want_idx) __builtin_amdgcn_raw_buffer_store_b32(pbytes, rs_p, off >> 2, 0, 0);
                    if (a.stats && a.stats_cf) {                         // statistics per (channel, output row)
                        s1 = wave_sum16(s1);
                        s2 = wave_sum16(s2);
                        if (lr == 0 && cv && orow_ok[fo_l]) {
                            double* dst = a.stats + ((size_t)slot * a.Cout * Fo + (size_t)cout * Fo + fo_of[fo_l]) * 2;
                            atomicAdd(dst, (double)s1);
                            atomicAdd(dst + 1, (double)s2);
                        }
                    }
                    c1 += s1; c2 += s2;
                }
            }
            if (a.stats && !a.stats_cf) {
                c1 = wave_sum16(c1);
                c2 = wave_sum16(c2);
                if (lr == 0 && cv) {
                    double* dst = a.stats + ((size_t)slot * a.Cout + cout) * 2;
                    atomicAdd(dst, (double)c1);
                    atomicAdd(dst + 1, (double)c2);
                }
            }
        }
        (void)FO_T;
        ct = ct_n; sp = sp_n; u_base = u_base_n;
        WX_TICK(pt_epi)
    }
#if WX_DBG & 128
    if (blockIdx.x == 0 && tid == 0) {
        a.y[0] = (float)(pt_mfma >> 10); a.y[1] = (float)(pt_bar >> 10); a.y[2] = (float)(pt_epi >> 10);
        a.y[3] = (float)((__builtin_amdgcn_s_memtime() - pt_begin) >> 10); a.y[4] = (float)nT; a.y[5] = (float)nChunks;
    }
#endif
}

template <bool POOL, bool DGRAD, bool UNPOOL, int CT = WX_CT>
static int launch_winox3(const ConvFwdArgs& a, hipStream_t s) {
    using C = WxCfg<POOL>;
    // the loaders address one clip with 32-bit byte offsets (buffer loads; 2^31 marks "out of range")
    if ((size_t)a.Cin * a.F * a.T * 4 >= (1ull << 30) || (size_t)a.Cout * a.F * a.T * 4 >= (1ull << 29)) {
        set_error("conv_winox3: one clip of the input / output must stay below 1 GiB / 512 MiB (Cin=%d Cout=%d F=%d T=%d)", a.Cin,
                  a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    if ((size_t)18 * a.CinP * a.CoutP * 6 >= (1ull << 31)) { set_error("conv_winox3: packed weights exceed 2 GiB"); return PBSED_E_ARG; }
    const int nTt = (a.T + WX_TT - 1) / WX_TT, nFt = (a.F + WX_FT - 1) / WX_FT;
    const int nSp = nTt * nFt * a.B, nCt = (a.Cout + CT - 1) / CT;          // CT = 32: the padding tiles of CoutP are never visited
    // item -> XCD map: 3 = the clips of a batch dealt over the XCDs (2 for fewer than 8 clips: spatial tiles dealt over them); the
    // other maps measured the same times (L2 locality is not what bounds the kernel)
    const int xcd_map = a.B < 8 ? 2 : 3;
    const int nWork = xcd_map == 3 ? (a.B + 7) / 8 * nTt * nFt * nCt * 8 : (nSp + 7) / 8 * 8 * nCt;
    if (a.T & 3) { set_error("conv_winox3: T = %d is not a multiple of 4 (rows must be 16-byte aligned; use the fp32 Winograd kernel)", a.T); return PBSED_E_UNSUPPORTED; }
    // persistent blocks, one per CU (110 KB of LDS each), a multiple of 8 so that an item's XCD is its block's XCD
    int blocks = device_cus() / 8 * 8;
    if (blocks < 8) blocks = 8;
    if (blocks > nWork) blocks = (nWork + 7) / 8 * 8;
    auto kern = conv_winox3_kernel<POOL, DGRAD, UNPOOL, CT>;
    PBSED_DYN_LDS_ONCE(kern, C::LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), C::LDS_BYTES, s, a, nCt, nSp, xcd_map, nWork);
    return check_launch("conv_winox3");
}

}  // namespace pbsed

using namespace pbsed;

static bool wx_ct32() { return true; }     // 32-cout blocks for launches that produce <= 32 channels (64: two idle consumer waves)

extern "C" {

void pbsed_conv_pack_dims_winox3(int Cin, int Cout, int dgrad, int* InP, int* OutP) {
    const int in = dgrad ? Cout : Cin, out = dgrad ? Cin : Cout;
    *InP = (in + WX_CK - 1) / WX_CK * WX_CK;
    *OutP = (out + WX_CT - 1) / WX_CT * WX_CT;
}

// up: uint16 [InP/32][xi 6][kh 3][OutP/16][part 3][lane 64][8]
int pbsed_pack_conv_weights_winox3(const float* w, unsigned short* up, int Cout, int Cin, int dgrad, void* stream) {
    int InP, OutP;
    pbsed_conv_pack_dims_winox3(Cin, Cout, dgrad, &InP, &OutP);
    const size_t total = (size_t)18 * InP * OutP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(winox3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, up, Cout, Cin, InP, OutP, dgrad);
    return check_launch("pack_conv_weights_winox3");
}

int pbsed_conv_fwd_winox3(const float* x, const unsigned short* u_packed, const float* bias, const float* scale,
                          const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx, double* stats,
                          int stats_per_cf, int B, int Cin, int Cout, int F, int T, int pool, void* stream) {
    if (pool && (F % 2)) { set_error("conv_fwd_winox3: pool needs even F"); return PBSED_E_ARG; }
    ConvFwdArgs a{};
    a.x = x; a.wp = reinterpret_cast<const float*>(u_packed); a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    pbsed_conv_pack_dims_winox3(Cin, Cout, 0, &a.CinP, &a.CoutP);
    if (Cout <= 32 && wx_ct32())
        return pool ? launch_winox3<true, false, false, 32>(a, (hipStream_t)stream) : launch_winox3<false, false, false, 32>(a, (hipStream_t)stream);
    return pool ? launch_winox3<true, false, false>(a, (hipStream_t)stream) : launch_winox3<false, false, false>(a, (hipStream_t)stream);
}

int pbsed_conv_bwd_data_winox3(const float* g, const unsigned short* ud_packed, const unsigned char* unpool_idx,
                               const int* seq_len, float* dz, const float* bx, const float* bmean, const float* binvstd,
                               const float* bscale, const float* bshift, int relu, double* stats, int B, int Cin, int Cout,
                               int F, int T, void* stream) {
    if (unpool_idx && (F % 2)) { set_error("conv_bwd_data_winox3: unpool needs even F"); return PBSED_E_ARG; }
    ConvFwdArgs a{};
    a.x = g; a.wp = reinterpret_cast<const float*>(ud_packed); a.seq_len = seq_len; a.y = dz; a.unpool_idx = unpool_idx;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = F; a.T = T;      // roles swapped
    pbsed_conv_pack_dims_winox3(Cin, Cout, 1, &a.CinP, &a.CoutP);
    if (Cin <= 32 && wx_ct32())                                  // the launch produces the layer's Cin channels
        return unpool_idx ? launch_winox3<false, true, true, 32>(a, (hipStream_t)stream) : launch_winox3<false, true, false, 32>(a, (hipStream_t)stream);
    return unpool_idx ? launch_winox3<false, true, true>(a, (hipStream_t)stream) : launch_winox3<false, true, false>(a, (hipStream_t)stream);
}

}  // extern "C"
