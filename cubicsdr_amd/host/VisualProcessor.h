// VisualProcessor.h -- input queue -> process() -> distribute to N output queues; API of the reference's
// src/process/VisualProcessor.h:14-155 (own implementation).
#pragma once
#include <algorithm>
#include <memory>
#include <mutex>
#include <vector>

#include "ThreadBlockingQueue.h"

template <typename InputDataType, typename OutputDataType>
class VisualProcessor {
public:
    typedef std::shared_ptr<InputDataType> InputDataTypePtr;
    typedef std::shared_ptr<OutputDataType> OutputDataTypePtr;
    typedef ThreadBlockingQueue<InputDataTypePtr> VisualInputQueueType;
    typedef ThreadBlockingQueue<OutputDataTypePtr> VisualOutputQueueType;
    typedef std::shared_ptr<VisualInputQueueType> VisualInputQueueTypePtr;
    typedef std::shared_ptr<VisualOutputQueueType> VisualOutputQueueTypePtr;

    virtual ~VisualProcessor() = default;

    bool isInputEmpty() { std::lock_guard<std::mutex> g(busy_update); return input ? input->empty() : true; }
    // "output empty" in the reference means: no attached queue is FULL (VisualProcessor.h:39-48)
    bool isOutputEmpty() {
        std::lock_guard<std::mutex> g(busy_update);
        for (auto &o : outputs) if (o->full()) return false;
        return true;
    }
    bool isAnyOutputEmpty() {
        std::lock_guard<std::mutex> g(busy_update);
        for (auto &o : outputs) if (!o->full()) return true;
        return false;
    }
    void setInput(VisualInputQueueTypePtr vis_in) { std::lock_guard<std::mutex> g(busy_update); input = vis_in; }
    void attachOutput(VisualOutputQueueTypePtr vis_out) { std::lock_guard<std::mutex> g(busy_update); outputs.push_back(vis_out); }
    void removeOutput(VisualOutputQueueTypePtr vis_out) {
        std::lock_guard<std::mutex> g(busy_update);
        outputs.erase(std::remove(outputs.begin(), outputs.end(), vis_out), outputs.end());
    }
    void flushQueues() {
        VisualInputQueueTypePtr in;
        std::vector<VisualOutputQueueTypePtr> outs;
        { std::lock_guard<std::mutex> g(busy_update); in = input; outs = outputs; }
        if (in) in->flush();
        for (auto &o : outs) o->flush();
    }
    void run() {
        VisualInputQueueTypePtr in;
        { std::lock_guard<std::mutex> g(busy_update); in = input; }
        if (in && !in->empty()) process();
    }

protected:
    virtual void process() = 0;
    // blocking (timeout 0) or timed push to every attached queue (VisualProcessor.h:132-145)
    void distribute(OutputDataTypePtr item, std::uint64_t timeout = BLOCKING_INFINITE_TIMEOUT, const char *errorMessage = nullptr) {
        std::lock_guard<std::mutex> g(busy_update);
        for (auto &o : outputs) (void)o->push(item, timeout, errorMessage);
    }

    VisualInputQueueTypePtr input;
    std::vector<VisualOutputQueueTypePtr> outputs;
    std::mutex busy_update;
};

// 1-to-n dispatchers without processing (reference VisualProcessor.h:164-224): they drain the input while at least one
// output has room and stop at the first item that would find every output full (that item is consumed, as in the reference).
#include <string>
#include <typeinfo>

#include "IOThread.h"

// every output receives the SAME instance (pointer re-dispatch)
template <typename OutputDataType>
class VisualDataDistributor : public VisualProcessor<OutputDataType, OutputDataType> {
    typedef VisualProcessor<OutputDataType, OutputDataType> Base;

protected:
    void process() override {
        typename Base::OutputDataTypePtr inp;
        typename Base::VisualInputQueueTypePtr in;
        { std::lock_guard<std::mutex> g(Base::busy_update); in = Base::input; }
        while (in && in->try_pop(inp)) {
            if (!Base::isAnyOutputEmpty()) return;         // do not distribute when all outputs are full
            if (inp) Base::distribute(inp);
        }
    }
};

// every item is deep-copied once into a pooled buffer, and that copy goes to all outputs
template <typename OutputDataType>
class VisualDataReDistributor : public VisualProcessor<OutputDataType, OutputDataType> {
    typedef VisualProcessor<OutputDataType, OutputDataType> Base;

public:
    VisualDataReDistributor() : buffers(std::string(typeid(*this).name())) {}

protected:
    ReBuffer<OutputDataType> buffers;
    void process() override {
        typename Base::OutputDataTypePtr inp;
        typename Base::VisualInputQueueTypePtr in;
        { std::lock_guard<std::mutex> g(Base::busy_update); in = Base::input; }
        while (in && in->try_pop(inp)) {
            if (!Base::isAnyOutputEmpty()) return;
            if (inp) {
                typename Base::OutputDataTypePtr outp = buffers.getBuffer();
                (*outp) = (*inp);
                Base::distribute(outp);
            }
        }
    }
};
