// hostpool.h -- the host threads of the library's plain-C helpers (packets.hip second pass, crc32.hip).  Creating std::threads per
// call cost more than the work of one export batch (7 of 10 ms for 3 x 10^4 association rows), so the threads are made once and
// parked on a condition variable.  One job at a time (a second caller waits); the pool is remade in a forked child (threads do not
// survive fork) and is deliberately never destroyed (its workers are detached and parked when the process ends).
#pragma once
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

class HostPool {
 public:
  // f(0) .. f(n_parts - 1), part 0 on the calling thread; returns when all are done.  n_parts is cut to the pool's size + 1.
  static void run(int n_parts, const std::function<void(int)>& f) {
    if (n_parts <= 1) { f(0); return; }
    HostPool* p = get();
    std::lock_guard<std::mutex> one_job(p->run_m_);
    n_parts = std::min(n_parts, p->n_workers_ + 1);
    {
      std::lock_guard<std::mutex> g(p->m_);
      p->job_ = &f;
      p->n_parts_ = n_parts;
      p->remaining_ = n_parts - 1;
      p->gen_++;
    }
    p->cv_job_.notify_all();
    f(0);
    std::unique_lock<std::mutex> g(p->m_);
    p->cv_done_.wait(g, [&] { return p->remaining_ == 0; });
    p->job_ = nullptr;
  }
  static int size() { return get()->n_workers_ + 1; }
  // parts a job of `n_items` should be cut into when one part should hold at least `grain` items
  static int parts_for(size_t n_items, size_t grain) { return (int)std::max<size_t>(1, std::min<size_t>((size_t)size(), n_items / std::max<size_t>(grain, 1))); }

 private:
  static HostPool* get() {
    static std::mutex make_m;
    static HostPool* pool = nullptr;
    std::lock_guard<std::mutex> g(make_m);
    if (!pool || pool->pid_ != getpid()) pool = new HostPool();      // (a forked child: the parent's pool object is abandoned)
    return pool;
  }
  HostPool() : pid_(getpid()) {
    const int hw = (int)std::thread::hardware_concurrency();
    n_workers_ = std::max(0, std::min(hw > 0 ? hw : 1, 16) - 1);
    for (int w = 0; w < n_workers_; w++) std::thread([this, w] { loop(w + 1); }).detach();
  }
  void loop(int part) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)>* f;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_job_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (part >= n_parts_) continue;
        f = job_;
      }
      (*f)(part);
      {
        std::lock_guard<std::mutex> g(m_);
        if (--remaining_ == 0) cv_done_.notify_all();
      }
    }
  }
  const pid_t pid_;
  int n_workers_ = 0;
  std::mutex run_m_, m_;
  std::condition_variable cv_job_, cv_done_;
  const std::function<void(int)>* job_ = nullptr;
  int n_parts_ = 0, remaining_ = 0;
  unsigned long long gen_ = 0;
};
