// vaa_rows.h — row-statistics records of K3's ROWS path, shared by vaa_loss.hip (K3) and vaa_head.hip (LM head fused with K3's statistics).
#pragma once
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

int rows_split(int R, int V);  // parts per row of the K3 workspace layout [R][4] PartStat + [R] SliceStat (vaa_loss.hip)

}  // namespace vaa
