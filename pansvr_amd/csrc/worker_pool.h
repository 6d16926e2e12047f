// worker_pool.h -- host worker threads that outlive a call (used by the command's reader / formatter / writer stages)
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

namespace psvr {

// Workers that live as long as the thread that uses them.  The stages of the command call for their `-t` threads several times per piece
// of a batch, a millisecond of work per thread each time: started anew per call that was 48 pthread_create + join in a row per call
// (~0.7 ms, a third of a stage's time per piece, and CPU time out of a container's quota).  Idle workers sleep on a condition variable.
class WorkerPool {
	std::mutex mu_;
	std::condition_variable go_, done_;
	std::vector<std::thread> th_;
	void (*call_)(void *, int) = nullptr;
	void *arg_ = nullptr;
	unsigned long long gen_ = 0;
	int want_ = 0, left_ = 0;
	bool stop_ = false;
	bool busy_ = false;                               // (only the owning thread looks at it)
	void loop(int id)
	{
		unsigned long long seen = 0;
		std::unique_lock<std::mutex> lk(mu_);
		for (;;) {
			go_.wait(lk, [&] { return stop_ || gen_ != seen; });
			if (stop_) return;
			seen = gen_;
			if (id > want_) continue;                     // (this call uses fewer workers than there are)
			void (*const call)(void *, int) = call_;
			void *const arg = arg_;
			lk.unlock();
			call(arg, id);
			lk.lock();
			if (--left_ == 0) done_.notify_one();
		}
	}
public:
	~WorkerPool()
	{
		{ std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
		go_.notify_all();
		for (std::thread &t : th_) t.join();
	}
	// fn(t) for t = 0 .. n-1: t = 0 on the calling thread, the others on workers 1 .. n-1; returns when all are through
	template <class F> void run(int n, F &&fn)
	{
		if (n <= 1) { if (n == 1) fn(0); return; }
		if (busy_) { for (int t = 0; t < n; ++t) fn(t); return; }      // called from inside its own fn(0): one after the other, here
		struct Busy { bool &b; explicit Busy(bool &x) : b(x) { b = true; } ~Busy() { b = false; } } mark(busy_);
		while ((int)th_.size() < n - 1) { const int id = (int)th_.size() + 1; th_.emplace_back([this, id] { loop(id); }); }
		{
			std::lock_guard<std::mutex> lk(mu_);
			call_ = [](void *a, int t) { (*(typename std::remove_reference<F>::type *)a)(t); };
			arg_ = (void *)&fn, want_ = n - 1, left_ = n - 1, ++gen_;
		}
		go_.notify_all();
		fn(0);
		std::unique_lock<std::mutex> lk(mu_);
		done_.wait(lk, [&] { return left_ == 0; });
	}
};
inline WorkerPool &thread_pool() { static thread_local WorkerPool p; return p; }       // one per calling thread (reader, formatter, writer ...)

} // namespace psvr
