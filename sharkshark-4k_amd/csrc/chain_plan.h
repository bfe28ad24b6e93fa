// Dependency plan of one layer of a conv chain (conv_chain.hip): pure host logic, no HIP - also compiled by the CPU test
// tests/test_chain_plan_cpu.py.
//
// Units of the chain are (item, tile); a tile's counter counts finished units.  For layer k (0-based in the chain) with
//   cum_k   = units per tile of all layers < k,      cum_km1 = units per tile of all layers < k - 1,
// a unit of layer k waits for `need0` on its 3 x 3 tile neighbourhood before its first K-chunk is read and for `need_new` before
// chunk `newest` is read (and before anything is written).  newest == 0: everything is waited for up front (need0 = need_new).
#pragma once
#include <cstddef>
#include <vector>

namespace ss4k {

struct ChainPrevLayer {
  const char* out_lo; const char* out_hi;   // address range of the planes the previous chain layer wrote
  int nitems;                               // its cout groups (units per tile)
};
struct ChainLayerPlan { int newest; unsigned need_old, need_new; };

// chunk_planes: the plane read by every K-chunk of this layer, in K order; plane_bytes: bytes of one plane of the previous
// layer's output tensor.  prev == nullptr: first layer of the chain (its inputs were complete before the launch).
inline ChainLayerPlan chain_plan_layer(const std::vector<const char*>& chunk_planes, const ChainPrevLayer* prev, size_t plane_bytes,
                                       unsigned cum_k, unsigned cum_km1) {
  ChainLayerPlan p{0, cum_km1, cum_k};
  if (!prev) return p;
  const int nch = (int)chunk_planes.size();
  auto is_new = [&](int c) { return chunk_planes[c] >= prev->out_lo && chunk_planes[c] < prev->out_hi; };
  // K-chunks whose planes the PREVIOUS layer wrote must form the tail of the K loop (the dense block's newest growth planes do)
  // and start at chunk 2 or later (the poll sits two chunks ahead of the DMA it guards); anything else is polled for up front
  int first = nch;
  for (int c = nch - 1; c >= 0 && is_new(c); --c) first = c;
  bool suffix_only = true;
  for (int c = 0; c < first; ++c) suffix_only = suffix_only && !is_new(c);
  p.newest = (suffix_only && first >= 2 && first < nch) ? first : 0;
  // conv1 of an RDB: EVERY chunk is the previous layer's (conv5's) output, which came from two units per tile - planes 0-1 from
  // the first cout group, planes 2-3 from the second, published in that order (ChainItem.pub_need).  Chunks 0-1 then only wait
  // for the first group's units (one unit short of the whole previous layer), chunks 2-3 for the second's: the unit starts
  // without blocking and the wait for the second group hides under the first two chunks
  if (prev->nitems == 2 && nch == 4 && is_new(0) && chunk_planes[0] == prev->out_lo && chunk_planes[1] == prev->out_lo + plane_bytes &&
      chunk_planes[2] == prev->out_lo + 2 * plane_bytes && chunk_planes[3] == prev->out_lo + 3 * plane_bytes) {
    p.newest = 2; p.need_old = cum_k - 1; p.need_new = cum_k;
  }
  return p;
}

}  // namespace ss4k
