// Stress of the engine's worker pool (hso_amd/host/hso_engine_impl.h): two pools on two caller threads, phases of 2..121 items of
// almost no work back to back — every item of every phase must run exactly once (a worker that is late leaving one phase must not
// take an index of the next), and no phase may hang.  Built and run by tests/test_engine_cpu.py.
#include "../hso_amd/host/hso_engine_impl.h"
#include <cstdio>
using namespace hso::engine;

int main(int argc, char** argv)
{
  const int n_threads = argc > 1 ? atoi(argv[1]) : 6, phases = argc > 2 ? atoi(argv[2]) : 100000;
  std::atomic<int> bad{0};
  auto caller = [&](int id) {
    Pool pool(n_threads);
    std::vector<std::atomic<int>> hit(128);
    for (int it = 0; it < phases && !bad.load(); it++) {
      const int n = 2 + (it * 7) % 120;
      for (int i = 0; i < n; i++) hit[i].store(0);
      pool.run(n, [&](int i) { hit[i].fetch_add(1); });
      for (int i = 0; i < n; i++)
        if (hit[i].load() != 1) { printf("pool %d phase %d: item %d ran %d times\n", id, it, i, hit[i].load()); bad.store(1); break; }
      if ((it % 1000) == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));   // lets the workers fall asleep now and then
    }
  };
  std::thread a(caller, 0), b(caller, 1);
  a.join(); b.join();
  if (!bad.load()) printf("ok\n");
  return bad.load();
}
