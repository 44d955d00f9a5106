#include <thread>
#include <vector>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
template <typename F>
int run_pool(int64_t n_tasks, int32_t n_threads, F&& task) {
    std::vector<int> status((size_t)n_tasks, 0);
    int64_t workers = n_threads;
    if (workers > n_tasks) workers = n_tasks;
    std::atomic<int64_t> next{0};
    auto loop = [&]() { for (;;) { const int64_t i = next.fetch_add(1, std::memory_order_relaxed); if (i >= n_tasks) return; status[(size_t)i] = task(i); } };
    if (workers == 1) loop();
    else { std::vector<std::thread> pool; for (int64_t w = 1; w < workers; ++w) pool.emplace_back(loop); loop(); for (auto& th : pool) th.join(); }
    return 0;
}
int main() {
    for (int thr : {1, 8}) {
        auto t0 = std::chrono::steady_clock::now();
        volatile double sink = 0;
        run_pool(64, thr, [&](int64_t i) { double s = 0; for (int k = 0; k < 2000000; ++k) s += k * 1e-9 * (i + 1); sink = sink + s; return 0; });
        auto t1 = std::chrono::steady_clock::now();
        printf("%d threads: %.2f ms\n", thr, std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
}
