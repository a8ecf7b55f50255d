// Index maps of the persistent scans' exchange (gru_stack.hip) in ONE place, shared TEXTUALLY by the kernels and by a host-side
// checker: which workgroup plays which role (gru_granule_role.inc), which word of the workspace a gate thread publishes, which
// words a contraction lane polls.  The kernels expand the macros below (same token stream as the expressions they replaced: the
// device ISA of gru_stack.hip is byte-identical with and without this header); oracle/scan_protocol_check.cpp (test infrastructure,
// CPU, g++) compiles the same text and enumerates it over the shapes the launchers accept - it checks what a sanitizer would
// have to find at run time: every polled word has exactly one publisher; it is the word of the (chain, layer, step, batch row,
// hidden unit) the consumer's MFMA operand order assumes; a ring's workgroups share an XCD (the XCD-local exchange relies on it);
// every XCD can hold its share of the grid; nothing leaves the workspace.
// Reference op site: torch.nn.GRU inside padertorch's GRU wrapper, pb_sed/models/weak_label/crnn.py:61-67.
#pragma once
#include <stddef.h>

// The exchanged state arrays are TILE-MAJOR: [chain * nlayers + layer][T][batch tile][H / 16 producers][16 rows][16 units] words;
// per_cl = T * Bp * H words of one (chain, layer).
// NB the macro bodies are the kernels' original expressions token by token - arguments are NOT parenthesised on purpose (a
// parenthesis or a `* 1` changes the expression tree, and with it the instruction order of kernels whose timing was tuned on
// hardware): pass plain identifiers, and for TILE0 either `by` or `by * tiles` (the block's first 16-row batch tile; evaluated
// left to right in 64 bits behind the cast).  The word-index macros are SUMS WITHOUT outer parentheses, to be written behind
// `pointer +` only (the kernels add the terms to the pointer one after the other); the host side wraps them (gmap:: below).
// First word of ring block (TILE0, bx)'s own first tile at step 0:
#define PBSED_GM_RING_BASE(chain, nlayers, layer, per_cl, TILE0, H, bx) \
    (size_t)(chain * nlayers + layer) * per_cl + ((size_t)TILE0 * (H / 16) + bx) * 256
// ... and what gate thread `tid` adds for step t (and for tile nb of the block: PBSED_GM_RING_WORD_NB): row (tid >> 4) & 15, unit
// tid & 15 of the producer's 1 KB tile
#define PBSED_GM_RING_WORD(t, Bp, H, tid) (size_t)t * Bp * H + (tid & 255)
#define PBSED_GM_RING_WORD_NB(t, Bp, H, nb, tid) (size_t)t * Bp * H + (size_t)nb * (H / 16) * 256 + (tid & 255)
// Byte offset (step 0, first tile of the block) of the first 16-byte poll load of lane (lq, lr) of the contraction wave whose K
// range starts at hidden unit k0; load n of the lane is n * 1024 bytes on (the next producer's tile), a time step Bp * H * 4, the
// block's next batch tile 16 * H * 4.  The lane's four words of load n are units k0 + 16 n + 4 lq + {0..3} of batch row lr.
#define PBSED_GM_POLL_OFFSET0(cl_src, per_cl, TILE0, H, k0, lr, lq) \
    ((unsigned)(((size_t)cl_src * per_cl + (size_t)TILE0 * 16 * H + (k0 / 16) * 256 + lr * 16 + lq * 4) * 4))

#if !defined(__HIPCC__)
// ---------------------------------------------------------------- host side (the checker): the same text behind plain functions
namespace pbsed {
namespace gmap {
struct Role { int bx, by, chain, gid; bool idle; };
struct RoleArgs { int ring_xcd, nchains, nlayers, nby; };
inline Role role_1d(unsigned block, const RoleArgs& a) {
    struct { unsigned x; } blockIdx = {block};
    Role r;
    r.idle = false;
    {
#include "gru_granule_role.inc"
    }
    return r;
}
// blocks per XCD of the 1-D grid (gru_stack.hip::granule_xcd_grid)
inline int slots_1d(int nchains, int nlayers, int nby, int nj) {
    const int R = nchains * nlayers * nby, P = nchains * (nlayers - 1) * nby;
    return (R + 7) / 8 * nj + (P * nj + 7) / 8;
}
}  // namespace gmap
}  // namespace pbsed
#endif
