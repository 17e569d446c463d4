// pt_feeder.h -- one host thread per device, feeding that device's stream (pt_multi.cpp).
//
// A frame of the single-process multi-GPU form is, per device, ~a dozen HIP calls (set-device, the path-kernel launch,
// the resolve, the device's part of the film gather).  Issued by ONE host thread for device 0 .. n-1 in turn, the last
// device starts its share n enqueue-times after the first one and the enqueue times add up per frame (measured: tools/r04/
// multi_enqueue.py); issued by n threads they overlap, and the caller's thread is free to post the next frame at once.
//
// Feeder = n worker threads, each with a FIFO of jobs.  post(w, job) appends to worker w's queue and returns; jobs of ONE
// worker run in the order they were posted (a device's stream therefore sees frame k before frame k + 1), jobs of
// different workers run concurrently.  drain() blocks until every queue is empty and every job has returned, and reports
// the first failure since the last drain.  No HIP, no RCCL in here: plain C++ threads, testable on a CPU box
// (pt_debug_feeder_selftest).
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace ptfeed {

// A job returns 0 or a status code; on failure it leaves a message in `err`.
using Job = std::function<int(std::string& err)>;

class Feeder {
public:
    explicit Feeder(unsigned n_workers) {
        for (unsigned w = 0; w < n_workers; ++w) workers_.emplace_back(new Worker());
        for (unsigned w = 0; w < n_workers; ++w) workers_[w]->thread = std::thread([this, w] { run(w); });
    }
    ~Feeder() {
        for (auto& w : workers_) {
            { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
            w->cv.notify_all();
        }
        for (auto& w : workers_) if (w->thread.joinable()) w->thread.join();
    }
    Feeder(const Feeder&) = delete;
    Feeder& operator=(const Feeder&) = delete;

    unsigned size() const { return (unsigned)workers_.size(); }

    void post(unsigned worker, Job job) {
        Worker& w = *workers_[worker];
        { std::lock_guard<std::mutex> lk(w.mu); w.queue.push_back(std::move(job)); }
        w.cv.notify_all();
    }

    // Every job posted so far has run when this returns.  Status / message of the first job that failed since the last
    // drain (jobs after a failed one still run: each device's stream must stay consistent with the others').
    int drain(std::string* err = nullptr) {
        for (auto& wp : workers_) {
            Worker& w = *wp;
            std::unique_lock<std::mutex> lk(w.mu);
            w.idle_cv.wait(lk, [&] { return w.queue.empty() && !w.busy; });
        }
        std::lock_guard<std::mutex> lk(err_mu_);
        const int rc = first_rc_;
        if (err) *err = first_err_;
        first_rc_ = 0;
        first_err_.clear();
        return rc;
    }

private:
    struct Worker {
        std::thread thread;
        std::mutex mu;
        std::condition_variable cv, idle_cv;
        std::deque<Job> queue;
        bool busy = false, stop = false;
    };
    void run(unsigned index) {
        Worker& w = *workers_[index];
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(w.mu);
                w.cv.wait(lk, [&] { return w.stop || !w.queue.empty(); });
                if (w.queue.empty()) return;            // stop requested and nothing left to run
                job = std::move(w.queue.front());
                w.queue.pop_front();
                w.busy = true;
            }
            std::string err;
            const int rc = job(err);
            if (rc != 0) {
                std::lock_guard<std::mutex> lk(err_mu_);
                if (first_rc_ == 0) { first_rc_ = rc; first_err_ = err; }
            }
            {
                std::lock_guard<std::mutex> lk(w.mu);
                w.busy = false;
            }
            w.idle_cv.notify_all();
        }
    }
    std::vector<std::unique_ptr<Worker>> workers_;
    std::mutex err_mu_;
    int first_rc_ = 0;
    std::string first_err_;
};

}  // namespace ptfeed
