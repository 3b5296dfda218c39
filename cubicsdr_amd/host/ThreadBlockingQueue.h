// ThreadBlockingQueue.h -- bounded multi-producer/multi-consumer queue with the API and timeout semantics of the
// reference's src/util/ThreadBlockingQueue.h:32-229 (own implementation).
//   timeout == BLOCKING_INFINITE_TIMEOUT (0)   : wait forever
//   timeout <= NON_BLOCKING_TIMEOUT (100 us)   : behave like try_push / try_pop
//   otherwise                                  : wait at most `timeout` microseconds, return false on expiry
// Capacity defaults to 1 and can only be raised (reference :62-70).
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <deque>
#include <iostream>
#include <memory>
#include <mutex>

#define MIN_ITEM_NB (1)
#define NON_BLOCKING_TIMEOUT (100)
#define BLOCKING_INFINITE_TIMEOUT (0)

class ThreadQueueBase {
public:
    virtual ~ThreadQueueBase() = default;
};
typedef std::shared_ptr<ThreadQueueBase> ThreadQueueBasePtr;

template <typename T>
class ThreadBlockingQueue : public ThreadQueueBase {
public:
    ThreadBlockingQueue() = default;
    ThreadBlockingQueue(const ThreadBlockingQueue &) = delete;
    ThreadBlockingQueue &operator=(const ThreadBlockingQueue &) = delete;

    void set_max_num_items(unsigned int n) {
        std::lock_guard<std::mutex> g(mu_);
        if (n > cap_) { cap_ = n; not_full_.notify_all(); }
    }

    bool push(const T &item, std::uint64_t timeout = BLOCKING_INFINITE_TIMEOUT, const char *errorMessage = nullptr) {
        std::unique_lock<std::mutex> g(mu_);
        auto has_room = [this] { return q_.size() < cap_; };
        if (timeout == BLOCKING_INFINITE_TIMEOUT) not_full_.wait(g, has_room);
        else if (timeout <= NON_BLOCKING_TIMEOUT) { if (!has_room()) return false; }
        else if (!not_full_.wait_for(g, std::chrono::microseconds(timeout), has_room)) {
            if (errorMessage) std::cout << "WARNING: push() timed out after " << (timeout * 0.001) << " ms: " << errorMessage << std::endl;
            return false;
        }
        q_.push_back(item);
        not_empty_.notify_all();
        return true;
    }
    bool try_push(const T &item) {
        std::lock_guard<std::mutex> g(mu_);
        if (q_.size() >= cap_) return false;
        q_.push_back(item);
        not_empty_.notify_all();
        return true;
    }
    bool pop(T &item, std::uint64_t timeout = BLOCKING_INFINITE_TIMEOUT, const char *errorMessage = nullptr) {
        std::unique_lock<std::mutex> g(mu_);
        auto has_item = [this] { return !q_.empty(); };
        if (timeout == BLOCKING_INFINITE_TIMEOUT) not_empty_.wait(g, has_item);
        else if (timeout <= NON_BLOCKING_TIMEOUT) { if (!has_item()) return false; }
        else if (!not_empty_.wait_for(g, std::chrono::microseconds(timeout), has_item)) {
            if (errorMessage) std::cout << "WARNING: pop() timed out after " << (timeout * 0.001) << " ms: " << errorMessage << std::endl;
            return false;
        }
        item = q_.front();
        q_.pop_front();
        not_full_.notify_all();
        return true;
    }
    bool try_pop(T &item) {
        std::lock_guard<std::mutex> g(mu_);
        if (q_.empty()) return false;
        item = q_.front();
        q_.pop_front();
        not_full_.notify_all();
        return true;
    }
    std::size_t size() const { std::lock_guard<std::mutex> g(mu_); return q_.size(); }
    bool empty() const { std::lock_guard<std::mutex> g(mu_); return q_.empty(); }
    bool full() const { std::lock_guard<std::mutex> g(mu_); return q_.size() >= cap_; }
    void flush() {
        std::lock_guard<std::mutex> g(mu_);
        q_.clear();
        not_full_.notify_all();
    }

private:
    mutable std::mutex mu_;
    std::condition_variable not_empty_, not_full_;
    std::deque<T> q_;
    std::size_t cap_ = MIN_ITEM_NB;
};
