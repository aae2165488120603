// Launch programs: record the kernel launches of one pass through the C ABI, replay them with a handful of host calls.
//
// Why: a training step (BASELINE config 3; scripts/main.py:116-145,188-197 of the reference) is ~900 launches of 50-600 us.  Issued one
// by one from Python through ctypes they cost the host ~11 ms of a 15 ms step (profiles/r14l_train_final.txt) - every launch pays the
// interpreter, the ctypes marshalling of its views and the plan look-ups again although nothing about it changes from step to step: the
// plans own every activation, gradient and packed filter, so pointers, grids and arguments are the same every time.  HIP graphs are the
// textbook answer and were measured SLOWER than eager issue on ROCm 7.2 (ssm_amd/training.py: 17.8 vs 15.6 ms of host time for the ~900
// nodes).  A program is the other way out: while it records, every launch of the library (ssm::launch, ssm_common.h) is executed as
// usual AND appended as a node - host function, grid, block, dynamic LDS, a copy of the argument values, the SLOT of its stream - and
// ssm_program_run issues the nodes [first, last) again with plain hipLaunchKernel calls, ~2 us each, no Python in between.  Work that
// is not the library's (torch kernels, the RCCL buckets) stays on the Python side between two ranges of nodes (ssm_program_mark).
//
// Threads: one program records at a time per process, and only the launches of the thread that called ssm_program_begin belong to it -
// other host threads (the reference's DataParallel replicas, scripts/main.py:74-76) keep launching eagerly, unrecorded.
//
// Streams: a program knows up to 8 stream SLOTS (begin: the handles of the recording pass; run: the handles to replay on - normally
// the same).  Cross-stream ordering inside a pass is ssm_stream_wait(src, dst) - "dst waits for everything queued on src so far" -
// executed with an event of the library and recorded as a node with an event of its own.
#include "ssm_common.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace ssm {

struct Node {
    enum Kind : int { KERNEL = 0, MEMSET = 1, WAIT = 2 };
    int kind;
    int slot, slot2;          // stream slot (WAIT: src, dst)
    const void *fn;           // KERNEL: host function; MEMSET: destination
    dim3 grid, block;
    unsigned lds;
    unsigned arg0, nargs;     // KERNEL: range in Recorder::arg_off; MEMSET: value / unused
    size_t bytes;             // MEMSET
    hipEvent_t ev;            // WAIT
};

constexpr int kMaxArgs = 64;          // parameters per recorded kernel (the library's widest takes 20)

struct Recorder {
    std::mutex mu;
    std::vector<Node> nodes;
    std::vector<unsigned> arg_off;              // offset of each argument's copy inside `blob`
    std::vector<unsigned char> blob;
    hipStream_t streams[8];
    int n_streams = 0;
    bool recording = false;
    std::thread::id owner;                      // the thread that called ssm_program_begin: only ITS launches belong to the pass
    int bad_stream = 0;                         // launches seen on a stream that is not one of the slots

    int slot_of(hipStream_t st) const {
        for (int i = 0; i < n_streams; ++i)
            if (streams[i] == st) return i;
        return -1;
    }
};

std::atomic<Recorder *> g_recorder{nullptr};

void record_kernel(Recorder *r, const void *fn, dim3 grid, dim3 block, unsigned lds, hipStream_t st, void *const *args, const size_t *sizes,
                   const size_t *aligns, int n) {
    std::lock_guard<std::mutex> lock(r->mu);
    if (!r->recording || std::this_thread::get_id() != r->owner) return;          // (other host threads' launches are not part of this pass)
    const int slot = r->slot_of(st);
    if (slot < 0 || n > kMaxArgs) {          // (a kernel with more parameters than ssm_program_run's argument table: refused at program_end)
        ++r->bad_stream;
        return;
    }
    Node nd{};
    nd.kind = Node::KERNEL;
    nd.slot = slot;
    nd.fn = fn;
    nd.grid = grid, nd.block = block, nd.lds = lds;
    nd.arg0 = (unsigned)r->arg_off.size();
    nd.nargs = (unsigned)n;
    for (int i = 0; i < n; ++i) {
        const size_t al = aligns[i] < 16 ? 16 : aligns[i];
        size_t off = (r->blob.size() + al - 1) / al * al;
        r->blob.resize(off + sizes[i]);
        memcpy(r->blob.data() + off, args[i], sizes[i]);
        r->arg_off.push_back((unsigned)off);
    }
    r->nodes.push_back(nd);
}

void record_memset(Recorder *r, void *dst, int value, size_t bytes, hipStream_t st) {
    std::lock_guard<std::mutex> lock(r->mu);
    if (!r->recording || std::this_thread::get_id() != r->owner) return;
    const int slot = r->slot_of(st);
    if (slot < 0) {
        ++r->bad_stream;
        return;
    }
    Node nd{};
    nd.kind = Node::MEMSET;
    nd.slot = slot;
    nd.fn = dst;
    nd.arg0 = (unsigned)value;
    nd.bytes = bytes;
    r->nodes.push_back(nd);
}

namespace {
// events of the eager ssm_stream_wait: a small ring per thread (a wait refers to the record that precedes it: reuse is safe)
struct EventRing {
    hipEvent_t ev[16] = {};
    int dev[16];
    unsigned next = 0;
    hipEvent_t get() {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned i = next++ & 15;
        if (ev[i] && dev[i] != d) {
            (void)hipEventDestroy(ev[i]);
            ev[i] = nullptr;
        }
        if (!ev[i]) {
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
            dev[i] = d;
        }
        return ev[i];
    }
};
thread_local EventRing g_ring;
}  // namespace

}  // namespace ssm

using ssm::Node;
using ssm::Recorder;

extern "C" int ssm_program_create(void **handle) {
    SSM_REQUIRE(handle, "program_create: null pointer");
    *handle = new Recorder();
    return SSM_OK;
}

extern "C" int ssm_program_destroy(void *handle) {
    if (!handle) return SSM_OK;
    Recorder *r = static_cast<Recorder *>(handle);
    Recorder *expected = r;
    ssm::g_recorder.compare_exchange_strong(expected, nullptr);
    for (Node &nd : r->nodes)
        if (nd.kind == Node::WAIT && nd.ev) (void)hipEventDestroy(nd.ev);
    delete r;
    return SSM_OK;
}

extern "C" int ssm_program_begin(void *handle, void *const *streams, int n_streams) {
    SSM_REQUIRE(handle && streams && n_streams >= 1 && n_streams <= 8, "program_begin: handle / 1..8 streams");
    Recorder *r = static_cast<Recorder *>(handle);
    SSM_REQUIRE(r->nodes.empty() && !r->recording, "program_begin: the program already holds a recording");
    for (int i = 0; i < n_streams; ++i) r->streams[i] = (hipStream_t)streams[i];
    r->n_streams = n_streams;
    r->owner = std::this_thread::get_id();
    r->recording = true;
    Recorder *expected = nullptr;
    if (!ssm::g_recorder.compare_exchange_strong(expected, r)) {
        r->recording = false;
        ssm::set_error("program_begin: another program is recording (one recording per process at a time)");
        return SSM_E_ARG;
    }
    return SSM_OK;
}

extern "C" int ssm_program_mark(void *handle, int *n_nodes) {
    SSM_REQUIRE(handle && n_nodes, "program_mark: null pointer");
    Recorder *r = static_cast<Recorder *>(handle);
    std::lock_guard<std::mutex> lock(r->mu);
    *n_nodes = (int)r->nodes.size();
    return SSM_OK;
}

extern "C" int ssm_program_end(void *handle, int *n_nodes) {
    SSM_REQUIRE(handle, "program_end: null handle");
    Recorder *r = static_cast<Recorder *>(handle);
    Recorder *expected = r;
    ssm::g_recorder.compare_exchange_strong(expected, nullptr);
    std::lock_guard<std::mutex> lock(r->mu);
    r->recording = false;
    if (n_nodes) *n_nodes = (int)r->nodes.size();
    SSM_REQUIRE(r->bad_stream == 0, "program_end: %d launch(es) of the recorded pass ran on a stream that is not one of the program's %d slots (or took more than %d parameters)",
                r->bad_stream, r->n_streams, ssm::kMaxArgs);
    return SSM_OK;
}

extern "C" int ssm_program_run(void *handle, int first, int last, void *const *streams, int n_streams) {
    SSM_REQUIRE(handle && streams, "program_run: null pointer");
    Recorder *r = static_cast<Recorder *>(handle);
    SSM_REQUIRE(!r->recording, "program_run: the program is still recording");
    SSM_REQUIRE(n_streams == r->n_streams, "program_run: %d streams given, the program was recorded with %d", n_streams, r->n_streams);
    SSM_REQUIRE(first >= 0 && first <= last && last <= (int)r->nodes.size(), "program_run: range [%d, %d) outside the program's %d nodes", first, last,
                (int)r->nodes.size());
    void *argv[ssm::kMaxArgs];
    for (int i = first; i < last; ++i) {
        const Node &nd = r->nodes[i];
        hipError_t e = hipSuccess;
        if (nd.kind == Node::KERNEL) {
            for (unsigned a = 0; a < nd.nargs; ++a) argv[a] = r->blob.data() + r->arg_off[nd.arg0 + a];
            e = hipLaunchKernel(nd.fn, nd.grid, nd.block, argv, nd.lds, (hipStream_t)streams[nd.slot]);
        } else if (nd.kind == Node::MEMSET) {
            e = hipMemsetAsync(const_cast<void *>(nd.fn), (int)nd.arg0, nd.bytes, (hipStream_t)streams[nd.slot]);
        } else {
            e = hipEventRecord(nd.ev, (hipStream_t)streams[nd.slot]);
            if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)streams[nd.slot2], nd.ev, 0);
        }
        if (e != hipSuccess) {
            ssm::set_error("program_run: node %d (kind %d): %s", i, nd.kind, hipGetErrorString(e));
            return SSM_E_LAUNCH;
        }
    }
    return ssm::check_launch("ssm_program_run");
}

extern "C" int ssm_stream_wait(void *src_stream, void *dst_stream) {
    if (src_stream == dst_stream) return SSM_OK;
    hipEvent_t ev = nullptr;
    Recorder *r = ssm::g_recorder.load(std::memory_order_acquire);
    int s0 = -1, s1 = -1;
    if (r) {
        std::lock_guard<std::mutex> lock(r->mu);
        if (r->recording && std::this_thread::get_id() == r->owner) {
            s0 = r->slot_of((hipStream_t)src_stream), s1 = r->slot_of((hipStream_t)dst_stream);
            if (s0 < 0 || s1 < 0) {
                ++r->bad_stream;
                r = nullptr;
            }
        } else {
            r = nullptr;
        }
    }
    if (r) {          // a node of the program, with an event of its own (alive as long as the program)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    } else {
        ev = ssm::g_ring.get();
    }
    SSM_REQUIRE(ev, "stream_wait: cannot create an event");
    hipError_t e = hipEventRecord(ev, (hipStream_t)src_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)dst_stream, ev, 0);
    if (e != hipSuccess) {
        ssm::set_error("stream_wait: %s", hipGetErrorString(e));
        return SSM_E_LAUNCH;
    }
    if (r) {
        std::lock_guard<std::mutex> lock(r->mu);
        Node nd{};
        nd.kind = Node::WAIT;
        nd.slot = s0, nd.slot2 = s1;
        nd.ev = ev;
        r->nodes.push_back(nd);
    }
    return SSM_OK;
}
