// az_worker.h -- a host thread that runs the jobs it is handed, in order: group.hip binds one to every GPU of a
// group so that each GPU's launches are issued by one thread.  Plain C++ (no HIP), so that the CPU test program
// can drive it under ThreadSanitizer.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

namespace az_worker {

class Worker {
public:
    // last_error: where a failed job's message is read from, on the worker's own thread (az_last_error is per thread)
    explicit Worker(const char* (*last_error)(void)) : last_error_(last_error), th_([this] { loop(); }) {}
    ~Worker() {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
        cv_.notify_all();
        th_.join();
    }
    Worker(const Worker&) = delete;
    Worker& operator=(const Worker&) = delete;
    void post(std::function<int()> job) {
        { std::lock_guard<std::mutex> lk(mu_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    // returns the job's status; its error text comes back in `msg`
    int wait(std::string& msg) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !busy_; });
        msg = msg_;
        return rc_;
    }
private:
    void loop() {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return quit_ || (busy_ && job_); });
                if (quit_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            const int rc = job();
            const std::string m = rc < 0 && last_error_ ? last_error_() : "";
            { std::lock_guard<std::mutex> lk(mu_); rc_ = rc; msg_ = m; busy_ = false; }
            cv_.notify_all();
        }
    }
    const char* (*last_error_)(void);
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, quit_ = false;
    int rc_ = 0;
    std::string msg_;
    std::thread th_;                             // last member: the thread starts with everything above in place
};

}  // namespace az_worker
