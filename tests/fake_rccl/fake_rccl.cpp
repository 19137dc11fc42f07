// fake_rccl.cpp -- TEST INFRASTRUCTURE, never loaded by the product: a stand-in for the handful of RCCL entry points
// csrc/pbre_comm_impl.hpp binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommCount, ncclGroupStart / End, ncclSend /
// ncclRecv, ncclGetVersion, ncclGetErrorString), over POSIX shared memory between the processes of ONE machine.  Selected through
// PBRE_RCCL_LIB, it lets world > 1 of pbre_step_gather_device / pbre_scatter_actions_device EXECUTE where the real RCCL cannot: on the
// CPU lane emulation (host buffers), and with two ranks on the one GPU of the test boxes (RCCL refuses two ranks on one device).
//
// Semantics kept: point-to-point messages matched per (source, destination) pair in posting order; calls between ncclGroupStart and
// ncclGroupEnd are only recorded and executed at ncclGroupEnd (sends first, so two ranks that both send and receive cannot deadlock);
// stream order is kept the blunt way -- the stream is drained before a buffer is read and the copy into a receive buffer is complete
// before the call returns.  FAKE_RCCL_DEVICE=1: buffers are HIP device pointers (hipMemcpy through libamdhip64, dlopen'ed).
//
// Rendezvous: the 128-byte unique id carries the name of a control segment in /dev/shm created by ncclGetUniqueId; a message is a
// file /dev/shm/<name>.<src>.<dst>.<seq> written by the sender, announced by a per-pair counter in the control segment, read and
// unlinked by the receiver.  Waits time out (FAKE_RCCL_TIMEOUT_S, default 60) with ncclSystemError instead of hanging a test run.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr int MAXR = 16;
struct Control {
    std::atomic<int> arrived;
    std::atomic<int> left;
    std::atomic<uint64_t> sent[MAXR][MAXR];      // messages src -> dst published so far
};
struct Comm {
    std::string name;
    int rank = 0, world = 1;
    Control* ctl = nullptr;
    uint64_t got[MAXR] = {0};                    // messages received from each source
    uint64_t put[MAXR] = {0};                    // messages sent to each destination
};
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; void* stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double timeout_s() { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e ? atof(e) : 60.0; }
bool device_mode() { const char* e = getenv("FAKE_RCCL_DEVICE"); return e && e[0] == '1'; }

// ---- HIP, only in device mode
struct Hip {
    void* h = nullptr;
    int (*Memcpy)(void*, const void*, size_t, int) = nullptr;
    int (*StreamSynchronize)(void*) = nullptr;
    bool load() {
        if (h) return true;
        for (const char* p : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", "/opt/rocm/lib/libamdhip64.so"}) if ((h = dlopen(p, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return false;
        *(void**)&Memcpy = dlsym(h, "hipMemcpy"); *(void**)&StreamSynchronize = dlsym(h, "hipStreamSynchronize");
        return Memcpy && StreamSynchronize;
    }
} g_hip;

size_t type_size(int dt) { switch (dt) { case 0: case 1: return 1; case 6: case 9: return 2; case 2: case 3: case 7: return 4; default: return 8; } }

int run_send(const Op& op) {
    Comm* c = op.comm;
    std::vector<char> host;
    const void* src = op.buf;
    if (device_mode()) {
        if (!g_hip.load()) return 2;
        if (g_hip.StreamSynchronize(op.stream) != 0) return 1;      // everything enqueued before the send has written the buffer
        host.resize(op.bytes);
        if (g_hip.Memcpy(host.data(), op.buf, op.bytes, 2 /* hipMemcpyDeviceToHost */) != 0) return 1;
        src = host.data();
    }
    const uint64_t seq = c->put[op.peer]++;
    char path[256];
    snprintf(path, sizeof path, "/dev/shm/%s.%d.%d.%llu", c->name.c_str(), c->rank, op.peer, (unsigned long long)seq);
    FILE* f = fopen(path, "wb");
    if (!f) return 2;
    const size_t w = fwrite(src, 1, op.bytes, f);
    fclose(f);
    if (w != op.bytes) return 2;
    c->ctl->sent[c->rank][op.peer].fetch_add(1, std::memory_order_release);
    return 0;
}
int run_recv(const Op& op) {
    Comm* c = op.comm;
    const uint64_t seq = c->got[op.peer];
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->sent[op.peer][c->rank].load(std::memory_order_acquire) <= seq) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return 2;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    c->got[op.peer]++;
    char path[256];
    snprintf(path, sizeof path, "/dev/shm/%s.%d.%d.%llu", c->name.c_str(), op.peer, c->rank, (unsigned long long)seq);
    FILE* f = fopen(path, "rb");
    if (!f) return 2;
    std::vector<char> host(op.bytes);
    const size_t r = fread(host.data(), 1, op.bytes, f);
    fclose(f);
    unlink(path);
    if (r != op.bytes) return 3;                 // size mismatch between the matched send and receive
    if (device_mode()) {
        if (!g_hip.load()) return 2;
        if (g_hip.StreamSynchronize(op.stream) != 0) return 1;      // earlier readers / writers of the receive buffer on that stream
        if (g_hip.Memcpy(op.buf, host.data(), op.bytes, 1 /* hipMemcpyHostToDevice */) != 0) return 1;
    } else std::memcpy(op.buf, host.data(), op.bytes);
    return 0;
}
int flush() {
    int rc = 0;
    for (const Op& op : g_ops) if (op.send && rc == 0) rc = run_send(op);
    for (const Op& op : g_ops) if (!op.send && rc == 0) rc = run_recv(op);
    g_ops.clear();
    return rc;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetVersion(int* v) { if (v) *v = 99999; return 0; }       // (a version code no real RCCL reports: tests assert the shim is what ran)
const char* ncclGetErrorString(int r) {
    switch (r) { case 0: return "no error"; case 1: return "fake_rccl: HIP call failed"; case 2: return "fake_rccl: system error (shared memory / timeout)";
                 case 3: return "fake_rccl: message size mismatch"; default: return "fake_rccl: invalid argument"; }
}
int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return 4;
    std::memset(id, 0, sizeof *id);
    static std::atomic<int> counter{0};
    snprintf(id->internal, sizeof id->internal, "pbre_fake_rccl_%d_%d_%lld", (int)getpid(), counter++,
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    const int fd = shm_open((std::string("/") + id->internal).c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) return 2;
    close(fd);                                   // (zero-filled: all counters start at 0)
    return 0;
}
int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > MAXR || rank < 0 || rank >= world) return 4;
    id.internal[127] = 0;
    Comm* c = new Comm();
    c->name = id.internal; c->rank = rank; c->world = world;
    const int fd = shm_open((std::string("/") + c->name).c_str(), O_RDWR, 0600);
    if (fd < 0) { delete c; return 2; }
    c->ctl = (Control*)mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->ctl == MAP_FAILED) { delete c; return 2; }
    c->ctl->arrived.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->arrived.load() < world) {     // collective, like the real call
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { munmap(c->ctl, sizeof(Control)); delete c; return 2; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    *comm = c;
    return 0;
}
int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 4;
    if (c->ctl->left.fetch_add(1) + 1 == c->world) shm_unlink((std::string("/") + c->name).c_str());      // the last rank out removes the segment
    munmap(c->ctl, sizeof(Control));
    delete c;
    return 0;
}
int ncclCommCount(const void* comm, int* n) { if (!comm || !n) return 4; *n = ((const Comm*)comm)->world; return 0; }
int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() { if (g_depth <= 0) return 4; if (--g_depth == 0) return flush(); return 0; }
int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, void* stream) {
    Comm* c = (Comm*)comm;
    if (!c || !buf || peer < 0 || peer >= c->world) return 4;
    g_ops.push_back(Op{true, const_cast<void*>(buf), count * type_size(dt), peer, c, stream});
    return g_depth ? 0 : flush();
}
int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, void* stream) {
    Comm* c = (Comm*)comm;
    if (!c || !buf || peer < 0 || peer >= c->world) return 4;
    g_ops.push_back(Op{false, buf, count * type_size(dt), peer, c, stream});
    return g_depth ? 0 : flush();
}
}  // extern "C"
