// vaa_rows.h — row-statistics records of K3's ROWS path, shared by vaa_loss.hip (K3) and vaa_head.hip (LM head fused with K3's statistics).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace vaa {

constexpr int kA0 = 31744;  // first action token (UADA.py:384)
constexpr int kNA = 256;    // action bins

struct RowMap {  // one per labelled position, in (b,k) row-major order of labels[b,k+1] != -100 (vaa_loss_rowmap_build)
    int b, k, lab, ord;
};
struct PartStat {  // one per (row, part)
    float m, s;    // max and sum exp(z - m) over the part
    float zlab;    // logit of the label if it lies in this part, else -inf
    int amax;      // argmax over the part (global column index), lowest index on ties
};
struct SliceStat {  // one per row
    float alse, E;
    int pred;
    int pad;
};

int rows_split(int R, int V);

// head workspace of vaa_head_loss_rows_stats: [R][ceil(V / 128)] PartStat, then (256-byte aligned) the action-column logits [R][256] bf16
constexpr int kHeadCols = 128;  // vocabulary columns per workgroup of head_stats_kernel
inline size_t head_ws_align(size_t n) { return (n + 255) / 256 * 256; }
inline size_t head_ws_slice_offset(int R, int V) { return head_ws_align((size_t)R * ((V + kHeadCols - 1) / kHeadCols) * sizeof(PartStat)); }  // parts per row of the K3 workspace layout [R][4] PartStat + [R] SliceStat (vaa_loss.hip)

}  // namespace vaa
