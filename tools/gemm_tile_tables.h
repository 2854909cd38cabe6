// Developer tool header (round 3): host-side planner of explicit tile tables for the table hook of csrc/gemm_f16.h (GemmArgs::tiles).
// Used by tools/gemm_tab_bench.hip to measure tile-height / dispatch-order policies (profiles/r3_gemm_tile_tables.txt): none beats
// uniform 128-row tiles, so the product computes its tile walk arithmetically and carries no tables.
#pragma once
#include <algorithm>
#include <vector>
namespace tts {
// ------------------------------------------------------------------------------------------------------------------------------
// Planner. Rows are handed out in 16-row blocks; XCD x owns a contiguous range of blocks (its activation rows and the current
// n-chunk's weight rows stay in its L2, as in the one-tile kernels). Inside an XCD the range is cut into m-tiles whose heights
// follow a cyclic pattern `h[0..nh)` (blocks, 2..8); every n-chunk (cn column tiles) walks the m-tiles in list order with the
// chunk's column tiles innermost.
// ------------------------------------------------------------------------------------------------------------------------------
struct GemmPlanSpec {
  int nh = 1;
  int h[16] = {8};
  int cn = 0;     // column tiles per n-chunk (0: all)
  int order = 0;  // 0: m-tiles in row order; 1: tallest first (stable); 2: column tile outermost inside a chunk
};

struct GemmPlan {
  std::vector<int4> host; // [8][len]
  int len = 0, tiles = 0;
  int4 *dev = nullptr;
};

static inline void gemm_plan_build(GemmPlan &p, int M, int N, const GemmPlanSpec &sp) {
  const int nb = M / 16, NT = N / 128, cn = sp.cn > 0 ? std::min(sp.cn, NT) : NT;
  std::vector<std::vector<int4>> lists(8);
  for (int x = 0; x < 8; x++) {
    const int b0 = (int)((long long)nb * x / 8), b1 = (int)((long long)nb * (x + 1) / 8);
    std::vector<std::pair<int, int>> mt; // (first block, height)
    int b = b0, i = 0;
    while (b < b1) {
      int hh = std::min(sp.h[i % sp.nh], b1 - b);
      if (b1 - b - hh == 1) { if (hh < 8) hh += 1; else hh -= 1; } // never leave a 1-block tile behind (a wave would own no rows)
      mt.push_back({b, hh});
      b += hh; i++;
    }
    if (!mt.empty() && mt.back().second == 1 && mt.size() > 1) { // whole range of 1 block cannot happen for M >= 256; defensive merge
      mt[mt.size() - 2].second += 1; mt.pop_back();
    }
    if (sp.order == 1) std::stable_sort(mt.begin(), mt.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &c) { return a.second > c.second; });
    for (int c0 = 0; c0 < NT; c0 += cn) {
      const int c1 = std::min(NT, c0 + cn);
      if (sp.order == 2) {
        for (int c = c0; c < c1; c++)
          for (auto &t : mt) lists[x].push_back(make_int4(t.first * 16, t.second, c * 128, 0));
      } else {
        for (auto &t : mt)
          for (int c = c0; c < c1; c++) lists[x].push_back(make_int4(t.first * 16, t.second, c * 128, 0));
      }
    }
  }
  // order 3..6: WHICH tiles share a CU. On gfx950 the per-XCD workgroup indices j, j + 32, j + 64, j + 96 of the first round land on
  // one CU (tools trace: HW_ID per workgroup); the row-order walk above therefore puts four tiles of ONE column (shared W tile, four
  // different A tiles) on a CU. Re-deal each run of 128 uniform tiles so that a CU gets: 3 = one row tile x 4 columns (shared A),
  // 4 = 2 row tiles x 2 columns, 5 = 4 different rows AND columns (nothing shared), 6 = the same tiles as order 0 (control).
  if (sp.order >= 3 && sp.order <= 6 && cn >= 4 && cn % 4 == 0) {
    for (int x = 0; x < 8; x++) {
      auto &l = lists[x];
      std::vector<int4> out(l.size());
      size_t base = 0;
      for (; base + 128 <= l.size(); base += 128) {
        // the 128 tiles of this run, as 128 / cn row tiles x cn columns (row tile outer): tile (r, c) = l[base + r * cn + c]
        const int R = 128 / cn; // requires cn | 128
        std::vector<int4> grp; // 32 groups of 4, in CU order
        auto T = [&](int r, int c) { return l[base + (size_t)r * cn + c]; };
        if (sp.order == 3) { for (int r = 0; r < R; r++) for (int c = 0; c < cn; c += 4) for (int k = 0; k < 4; k++) grp.push_back(T(r, c + k)); }
        else if (sp.order == 4) { for (int r = 0; r < R; r += 2) for (int c = 0; c < cn; c += 2) { grp.push_back(T(r, c)); grp.push_back(T(r, c + 1)); grp.push_back(T(r + 1, c)); grp.push_back(T(r + 1, c + 1)); } }
        else if (sp.order == 5) { for (int r = 0; r < R; r += 4) for (int c = 0; c < cn; c++) for (int k = 0; k < 4; k++) grp.push_back(T(r + k, (c + k) % cn)); }
        else { for (int g = 0; g < 32; g++) for (int k = 0; k < 4; k++) grp.push_back(l[base + g + 32 * k]); }
        for (int g = 0; g < 32; g++) for (int k = 0; k < 4; k++) out[base + g + 32 * k] = grp[(size_t)g * 4 + k];
      }
      for (; base < l.size(); base++) out[base] = l[base];
      l.swap(out);
    }
  }
  p.len = 0; p.tiles = 0;
  for (auto &l : lists) { p.len = std::max(p.len, (int)l.size()); p.tiles += (int)l.size(); }
  p.host.assign((size_t)8 * p.len, make_int4(0, 0, 0, 0));
  for (int x = 0; x < 8; x++) std::copy(lists[x].begin(), lists[x].end(), p.host.begin() + (size_t)x * p.len);
}

} // namespace tts
