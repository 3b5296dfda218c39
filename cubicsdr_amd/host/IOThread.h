// IOThread.h -- thread base class with named queue bindings + ReBuffer recycling pool; API of the reference's
// src/IOThread.h:46-209 / IOThread.cpp:41-132 (own implementation, header-only).
#pragma once
#include <atomic>
#include <chrono>
#include <deque>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "ThreadBlockingQueue.h"

#define REBUFFER_GC_LIMIT 100

// Pool of shared_ptr<BufferType>: an entry is free for re-use when the pool holds the only reference
// (use_count() == 1, reference IOThread.h:62-129); unused entries age and the oldest is dropped after
// REBUFFER_GC_LIMIT misses.
template <typename BufferType>
class ReBuffer {
public:
    typedef std::shared_ptr<BufferType> ReBufferPtr;
    explicit ReBuffer(std::string bufferId) : id_(std::move(bufferId)) {}
    virtual ~ReBuffer() = default;

    ReBufferPtr getBuffer() {
        std::lock_guard<std::mutex> g(mu_);
        ReBufferPtr chosen;
        for (auto &e : pool_) {
            if (e.ptr.use_count() != 1) continue;          // still referenced by a consumer
            if (!chosen) { chosen = e.ptr; e.age = 1; }
            else --e.age;
        }
        if (chosen) {
            if (!pool_.empty() && pool_.back().age < -REBUFFER_GC_LIMIT) pool_.pop_back();
            return chosen;
        }
        pool_.push_back(Entry{std::make_shared<BufferType>(), 1});
        return pool_.back().ptr;
    }
    void purge() { std::lock_guard<std::mutex> g(mu_); pool_.clear(); }
    std::size_t size() const { std::lock_guard<std::mutex> g(mu_); return pool_.size(); }

private:
    struct Entry { ReBufferPtr ptr; int age; };
    std::string id_;
    std::deque<Entry> pool_;
    mutable std::mutex mu_;
};

class IOThread {
public:
    IOThread() : terminated(false), stopping(false) {}
    virtual ~IOThread() = default;

    // the thread entry point: run() until it returns; a throw marks the thread terminated and is re-thrown (IOThread.cpp:44-51)
    void threadMain() {
        terminated.store(false);
        stopping.store(false);
        try { run(); }
        catch (...) { terminated.store(true); stopping.store(true); throw; }
        terminated.store(true);
        stopping.store(true);
    }
    virtual void run() {}
    virtual void terminate() { stopping.store(true); }
    bool isStopping() { return stopping.load(); }
    bool isTerminated(int waitMs = 0) {                    // IOThread.cpp:101-132: poll in 1 ms steps up to waitMs
        if (terminated.load()) return true;
        for (int i = 0; i < waitMs && !terminated.load(); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        return terminated.load();
    }

    virtual void onBindOutput(std::string /*name*/, ThreadQueueBasePtr /*q*/) {}
    virtual void onBindInput(std::string /*name*/, ThreadQueueBasePtr /*q*/) {}
    void setInputQueue(const std::string &name, const ThreadQueueBasePtr &q) {
        { std::lock_guard<std::mutex> g(qmu_); inputs_[name] = q; }
        onBindInput(name, q);
    }
    ThreadQueueBasePtr getInputQueue(const std::string &name) {
        std::lock_guard<std::mutex> g(qmu_);
        auto it = inputs_.find(name);
        return it == inputs_.end() ? nullptr : it->second;
    }
    void setOutputQueue(const std::string &name, const ThreadQueueBasePtr &q) {
        { std::lock_guard<std::mutex> g(qmu_); outputs_[name] = q; }
        onBindOutput(name, q);
    }
    ThreadQueueBasePtr getOutputQueue(const std::string &name) {
        std::lock_guard<std::mutex> g(qmu_);
        auto it = outputs_.find(name);
        return it == outputs_.end() ? nullptr : it->second;
    }

protected:
    std::atomic_bool terminated, stopping;

private:
    std::mutex qmu_;
    std::map<std::string, ThreadQueueBasePtr> inputs_, outputs_;
};
