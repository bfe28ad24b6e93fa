// CPU harness of sharkshark-4k_amd/csrc/chain_plan.h (tools/dev_tests/test_chain_plan_cpu.py compiles and runs it): the dependency plan
// of three RDBs laid out the way Model::forward lays them out (trunk buffers rotating a -> t1 -> t2 -> a, growth planes reused).
#include "../../sharkshark-4k_amd/csrc/chain_plan.h"
#include <cstdio>
#include <cstdlib>
using namespace ss4k;
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main() {
  const size_t pb = 1 << 20;                       // bytes per plane
  static char X[3][4 << 20], G[8 << 20];           // three trunk tensors of 4 planes, growth tensor of 8 planes (addresses only)
  unsigned cum = 0, cum_prev = 0;
  ChainPrevLayer prev{}; bool have_prev = false;
  int cur = 0;
  for (int rdb = 0; rdb < 3; ++rdb) {
    const int nxt = (cur + 1) % 3;
    for (int c = 0; c < 5; ++c) {
      std::vector<const char*> planes;
      for (int q = 0; q < 4; ++q) planes.push_back(X[cur] + q * pb);
      for (int q = 0; q < 2 * c; ++q) planes.push_back(G + q * pb);
      const ChainLayerPlan p = chain_plan_layer(planes, have_prev ? &prev : nullptr, pb, cum, cum_prev);
      const int groups = c < 4 ? 1 : 2;
      if (!have_prev) {                            // first layer of the chain: nothing to wait for
        CHECK(p.newest == 0 && p.need_new == 0 && p.need_old == 0);
      } else if (c == 0) {                         // conv1: all four chunks are conv5's output -> split by conv5's two groups
        CHECK(p.newest == 2 && p.need_new == cum && p.need_old == cum - 1);
      } else {                                     // conv2..5: the newest growth planes are the last two chunks
        CHECK(p.newest == 4 + 2 * (c - 1) && p.need_new == cum && p.need_old == cum_prev);
        CHECK(p.newest >= 2 && p.newest == (int)planes.size() - 2);
      }
      CHECK(p.need_old <= p.need_new);
      char* out_lo = c < 4 ? G + 2 * c * pb : X[nxt];
      prev = ChainPrevLayer{out_lo, out_lo + (c < 4 ? 2 : 4) * pb, groups};
      have_prev = true;
      cum_prev = cum; cum += groups;
    }
    cur = nxt;
  }
  CHECK(cum == 3 * 6);
  // a layer whose new planes are NOT the tail of its K loop waits for everything up front
  {
    ChainPrevLayer pv{G, G + 2 * pb, 1};
    std::vector<const char*> planes = {G, G + pb, X[0], X[0] + pb};
    const ChainLayerPlan p = chain_plan_layer(planes, &pv, pb, 7, 6);
    CHECK(p.newest == 0 && p.need_new == 7);
    std::vector<const char*> one_late = {X[0], G, G + pb};          // would start at chunk 1: too early for the look-ahead
    const ChainLayerPlan p2 = chain_plan_layer(one_late, &pv, pb, 7, 6);
    CHECK(p2.newest == 0);
    std::vector<const char*> none = {X[0], X[0] + pb, X[0] + 2 * pb};   // reads nothing of the previous layer: still waits for it
    const ChainLayerPlan p3 = chain_plan_layer(none, &pv, pb, 7, 6);    // (the write-after-read argument needs the chain of waits)
    CHECK(p3.newest == 0 && p3.need_new == 7);
  }
  printf("chain plan ok\n");
  return 0;
}
