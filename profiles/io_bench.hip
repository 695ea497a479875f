// What the layout writer's host side can reach on a box: cost of pinning memory, device-to-host rate into pinned
// memory, and the rate at which T threads put bytes into files (pwrite into the page cache).
// Build: hipcc --offload-arch=gfx950 -O2 -o io_bench io_bench.hip -lpthread ; run: ./io_bench <dir> [total MiB]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const size_t total = (argc > 2 ? atoll(argv[2]) : 4096) << 20;
  const size_t piece = 16 << 20;
  CK(hipSetDevice(0));
  void* dev = nullptr;
  CK(hipMalloc(&dev, 1ull << 30));
  CK(hipMemset(dev, 0x5A, 1ull << 30));
  for (size_t mib : {64, 256, 1024}) {
    double t0 = now();
    void* p = nullptr;
    CK(hipHostMalloc(&p, mib << 20, hipHostMallocDefault));
    double t1 = now();
    CK(hipMemcpy(p, dev, mib << 20, hipMemcpyDeviceToHost));
    double t2 = now();
    CK(hipMemcpy(p, dev, mib << 20, hipMemcpyDeviceToHost));
    double t3 = now();
    printf("pin %4zu MiB: %.1f ms; first D2H %.1f GB/s, second %.1f GB/s\n", mib, (t1 - t0) * 1e3, (mib << 20) / (t2 - t1) / 1e9,
           (mib << 20) / (t3 - t2) / 1e9);
    CK(hipHostFree(p));
  }
  void* pin = nullptr;
  CK(hipHostMalloc(&pin, 1ull << 30, hipHostMallocDefault));
  CK(hipMemcpy(pin, dev, 1ull << 30, hipMemcpyDeviceToHost));
  for (int threads : {1, 2, 4, 8, 16, 32}) {
    for (int files_per_thread : {1, 64}) {
      std::atomic<size_t> next{0};
      const size_t pieces = total / piece;
      double t0 = now();
      std::vector<std::thread> pool;
      for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
          std::vector<int> fds;
          for (int f = 0; f < files_per_thread; ++f) {
            const std::string path = dir + "/io_bench_" + std::to_string(t) + "_" + std::to_string(f);
            fds.push_back(open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644));
          }
          std::vector<size_t> at(fds.size(), 0);
          for (size_t k = 0;; ++k) {
            const size_t i = next.fetch_add(1);
            if (i >= pieces) break;
            const int f = (int)(k % fds.size());
            const char* src = (const char*)pin + (i * piece) % (1ull << 30);
            if (pwrite(fds[f], src, piece, (off_t)at[f]) != (ssize_t)piece) { perror("pwrite"); break; }
            at[f] += piece;
          }
          for (int fd : fds) close(fd);
        });
      for (auto& th : pool) th.join();
      double t1 = now();
      printf("%2d threads x %2d files: %.2f GB in %.0f ms = %.2f GB/s\n", threads, files_per_thread, total / 1e9, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
      for (int t = 0; t < threads; ++t)
        for (int f = 0; f < files_per_thread; ++f) unlink((dir + "/io_bench_" + std::to_string(t) + "_" + std::to_string(f)).c_str());
    }
  }
  return 0;
}
