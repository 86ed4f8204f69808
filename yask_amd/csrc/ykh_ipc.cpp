// ykh_ipc.cpp -- halo transport between the GPUs (or GPU processes) of ONE host that needs no compute units to move the
// bytes: device-to-device copies into the neighbour's own buffers, mapped through HIP IPC memory handles, ordered by flag
// words in memory every device of the job sees coherently.  YASK_HIP_TRANSPORT=ipc (yk_env_init_from_launcher) or yk_env_init_ipc().
//
// Why (VERDICT r02 missing #4).  The reference progresses its MPI requests WHILE the interior is computed
// (StencilContext::adv_halo_exchange, src/kernel/lib/halo.cpp:494-574, called per micro-block, context.cpp:1037-1040).
// RCCL's ncclSend/ncclRecv are KERNELS: they need CUs, and a marching stencil launch keeps one 512-thread workgroup resident
// on every CU until it ends -- the "overlapped" exchange then starts when the interior is over.  A copy issued with
// hipMemcpyAsync() between two devices is executed by an SDMA engine over xGMI (blit kernels of a few waves where the runtime
// prefers them -- e.g. between two processes on the same device, which is how this file is tested on a one-GPU box); either
// way it does not queue behind the stencil's workgroups.
//
// Protocol.  Every rank owns a MAILBOX of 32-bit words, mapped into every other rank.  A directed link S -> R has up to CHANS
// CHANNELS; a channel is one (direction tag, buffer key) of R's receive buffers (HaloMsg::key names it alike on both ends) and
// is numbered by R the first time R posts a receive for it.  For exchange number e of a channel c:
//   R, on its comm stream, when its receive buffer may be overwritten:      S.mailbox[R][c].cts   = e      (set_words kernel)
//   S, on its comm stream: wait  S.mailbox[R][c].cts >= e                                                  (wait_words kernel)
//                          hipMemcpyAsync(R's buffer <- S's packed halo)                                   (copy engine / blit)
//                          R.mailbox[S][c].ready = e                                                       (set_words kernel)
//   R, on its comm stream: wait  R.mailbox[S][c].ready >= e, then unpack.
// Everything is stream-ordered: no host thread waits for the GPU, a run_solution() call queues all its steps.
//
// The host is NOT in the loop of an exchange (round 4): the address of a receive buffer -- IPC handle of
// its allocation + offset -- travels over the TCP mesh (ykh_launch.cpp) ONCE, when its channel is first used (a Registration,
// 96 bytes), and is cached by the sender; round 3 sent it with every message.  In steady state an exchange is kernel launches
// and copies only; yk_env_get_transport_counters() reports the control traffic so that a test can assert it stays flat.
// Cached addresses die with the buffers: exch_reset (prepare_solution(), free_halo_buffers(), Var::allocate / release /
// fuse_with) marks this rank's registrations stale, and exch_begin -- called by EVERY rank at the start of each
// run_solution() / exchange_halos() -- agrees on "somebody is stale" with one 8-byte all-reduce per call (not per step) and
// then resets all ranks together (drain, close mappings, zero the mailboxes, restart the epochs).
//
// Where the mailbox lives (flag words polled across devices must not sit in a cache).  In order of
// preference: uncached device memory (hipDeviceMallocUncached), fine-grained device memory, pinned HOST memory shared through
// POSIX shm and registered with every rank's device (system-coherent by construction; polls cross PCIe, ~1 us).  PLAIN
// hipMalloc memory is accepted only when every rank of the job sits on the SAME device (the one-GPU test set-up): ranks on
// different devices refuse it.  YASK_HIP_MAILBOX=uncached|finegrained|host|plain forces a kind (plain is still refused
// across devices).
// A waiter that is never released gives up after YASK_HIP_WAIT_TIMEOUT_S seconds (default 20) and raises the mailbox's error
// word; exch_check() (called by run_solution() once the streams have drained) turns that into an exception and prints the
// state of every channel (epoch expected / word found) instead of leaving a hung box.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/yask_hip_c_api.h"
#include "ykh_handles.hpp"
#include "ykh_mesh.hpp"
#include "ykh_runtime.hpp"

namespace {
using namespace ykh_mesh;

constexpr int TAGS = 27;                  // direction tags (3^3)
constexpr int CHANS = 256;                // channels of one directed link (tags x buffers: 26 packed + 2 per var slot in x)
constexpr int WORDS_PER_PEER = CHANS * 2;

struct Registration {         // "my receive buffer for (tag, key) of your messages is channel `chan` and lives here"
    int tag, key, chan;
    unsigned pad;
    hipIpcMemHandle_t handle;
    unsigned long long offset, room;      // room: bytes from there to the end of the allocation
};
struct SendChan { int chan; void* dst; unsigned long long room; unsigned epoch; };
struct RecvChan { int chan; void* buf; unsigned epoch; };
struct MailboxAd {            // how to map a rank's mailbox: an IPC handle (device memory) or a shm name (host memory)
    int kind;                 // 0 uncached, 1 fine-grained, 2 plain, 3 host
    hipIpcMemHandle_t handle;
    char shm[64];
    char busid[64];           // PCI bus id of the rank's device
};
const char* const KIND_NAME[] = {"uncached device", "fine-grained device", "plain device", "pinned host (shm)"};

struct IpcState {
    TcpState* mesh = nullptr;
    int rank = 0, nranks = 1;
    bool multi_device = false;                   // some rank of the job sits on another device than this one
    int kind = -1;
    unsigned* mailbox = nullptr;                 // my words + one error word at the end (device-visible address)
    void* mailbox_host = nullptr;                // kind 3: the mmap()ed shm segment behind it
    std::string shm_name;
    size_t mailbox_words = 0;
    std::vector<unsigned*> peer_mailbox;         // other ranks' mailboxes, mapped here (own entry = mailbox)
    std::vector<void*> peer_host;                // kind 3: their mmap()ed segments
    std::vector<int> peer_kind;
    std::map<std::string, void*> opened;         // "<peer>:<handle bytes>" -> base address of the mapping
    std::map<void*, hipIpcMemHandle_t> exported; // base of one of MY allocations -> its handle
    std::map<long long, SendChan> send_chan;     // (peer, tag, key) -> where my message goes
    std::map<long long, RecvChan> recv_chan;     // (peer, the SENDER's tag, key) -> which channel I gave it
    std::vector<int> next_chan;                  // per sender: channels handed out so far
    int verbose = 0;                             // YASK_HIP_IPC_VERBOSE=2: every control-plane step on stderr
    bool stale = false;                          // buffers behind registrations were freed / moved since the last collective reset
    double timeout_s = 20.0;
    std::vector<const unsigned*> wait_ptr;       // per message of the exchange in flight: its `ready` word ...
    std::vector<unsigned> wait_val;              // ... and the epoch it must reach
    long long ctl_msgs = 0, ctl_bytes = 0, begins = 0, resets = 0, dev_ops = 0;
    double open_limit_s() const { return timeout_s * 2 > 20.0 ? timeout_s * 2 : 20.0; }      // mapping a peer's allocation: see open_guarded()
    unsigned* err() const { return mailbox + mailbox_words - 1; }
    unsigned* word(unsigned* box, int peer, int chan, int kind) const { return box + ((size_t)peer * CHANS + chan) * 2 + kind; }
};
long long chan_key(int peer, int tag, int key) { return (((long long)peer * TAGS + tag) << 32) | (unsigned)key; }

// IPC handle of the allocation `p` lies in, p's offset within it and the bytes from p to its end
bool export_buf(IpcState* st, void* p, hipIpcMemHandle_t* h, unsigned long long* off, unsigned long long* room) {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return false; }
    auto it = st->exported.find((void*)base);
    if (it == st->exported.end()) {
        hipIpcMemHandle_t nh;
        if (hipIpcGetMemHandle(&nh, (void*)base) != hipSuccess) { (void)hipGetLastError(); return false; }
        it = st->exported.emplace((void*)base, nh).first;
    }
    *h = it->second;
    *off = (unsigned long long)((char*)p - (char*)base);
    *room = (unsigned long long)size - *off;
    return true;
}
// hipIpcOpenMemHandle() with a time limit.  The call has been seen to NEVER return (two processes mapping each other's multi-GB
// allocations at the same instant; tools/microbench/ipc_open.hip is the stand-alone probe): it runs on a helper
// thread, and a caller that has waited `limit_s` gets hipErrorNotReady -- an exception in the host instead of a hung job.  A normal
// open costs one thread start (~50 us, first use of a channel only).
// After a time-out: the helper is still INSIDE the HIP runtime.  The transport is marked wedged for the life of the process
// (`ipc_wedged`): ipc_free() then neither synchronises the device nor closes mappings (either may block behind the stuck call, which
// is the hang the time limit was there to prevent), every later open fails at once, and a helper that does come back late closes the
// mapping it got instead of leaking it.  A host that sees this error should leave through _exit(): normal teardown would run the HIP
// runtime's exit handlers with a thread still inside it.
std::atomic<bool> ipc_wedged{false};
hipError_t open_guarded(void** base, const hipIpcMemHandle_t& h, double limit_s) {
    struct Job { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; hipError_t rc = hipErrorUnknown; void* base = nullptr; hipIpcMemHandle_t h; int dev = 0; };
    if (ipc_wedged.load()) return hipErrorNotReady;
    auto job = std::make_shared<Job>();
    job->h = h;
    (void)hipGetDevice(&job->dev);
    std::thread([job] {
        void* b = nullptr;
        hipError_t rc = hipSetDevice(job->dev);
        if (rc == hipSuccess) rc = hipIpcOpenMemHandle(&b, job->h, hipIpcMemLazyEnablePeerAccess);
        if (rc != hipSuccess) (void)hipGetLastError();
        bool late;
        {
            std::lock_guard<std::mutex> g(job->m);
            job->rc = rc; job->base = b; job->done = true;
            late = job->abandoned;
            job->cv.notify_all();
        }
        if (late && rc == hipSuccess && b) (void)hipIpcCloseMemHandle(b);      // nobody will ever record this mapping: give it back
    }).detach();
    std::unique_lock<std::mutex> lk(job->m);
    if (!job->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return job->done; })) {
        job->abandoned = true;
        ipc_wedged.store(true);
        return hipErrorNotReady;
    }
    *base = job->base;
    return job->rc;
}
const char* open_error(hipError_t rc) {
    return rc == hipErrorNotReady ? "hipIpcOpenMemHandle did not return within the time limit (YASK_HIP_WAIT_TIMEOUT_S); the IPC transport is disabled for "
                                    "the rest of this process, which should leave through _exit()" : hipGetErrorString(rc);
}

void* import_buf(IpcState* st, int peer, const hipIpcMemHandle_t& h) {
    std::string key = std::to_string(peer) + ":" + std::string((const char*)&h, sizeof(h));
    auto it = st->opened.find(key);
    if (it != st->opened.end()) return it->second;
    void* base = nullptr;
    const hipError_t rc = open_guarded(&base, h, st->open_limit_s());
    if (rc != hipSuccess) {
        fprintf(stderr, "yask ipc transport: rank %d cannot map a buffer of rank %d: %s\n", st->rank, peer, open_error(rc));
        return nullptr;
    }
    st->opened.emplace(key, base);
    return base;
}
// the buffers behind my registrations are about to be (or have been) freed or moved: nothing cached may be used again.  Local and
// cheap (called per var by prepare_solution()); the ranks act on it together in ipc_begin().
void ipc_reset(void* user) {
    IpcState* st = static_cast<IpcState*>(user);
    st->exported.clear();
    st->stale = true;
}
// COLLECTIVE (every rank, start of every run_solution() / exchange_halos()): one 8-byte all-reduce; when any rank is stale,
// all ranks drop every registration together.
int ipc_begin(void* user) {
    IpcState* st = static_cast<IpcState*>(user);
    st->begins++;
    long long v = st->stale ? 1 : 0;
    if (st->verbose > 1) fprintf(stderr, "ipc[%d]: begin #%lld (stale %lld)\n", st->rank, st->begins, v);
    if (tcp_allreduce(st->mesh, 2, &v) != 0) return 1;
    if (st->verbose > 1) fprintf(stderr, "ipc[%d]: begin agreed: %s\n", st->rank, v ? "reset" : "keep");
    if (!v) return 0;
    st->resets++;
    if (hipDeviceSynchronize() != hipSuccess) return 1;       // my copies and flag stores of earlier exchanges have landed
    long long z = 0;
    if (tcp_allreduce(st->mesh, 0, &z) != 0) return 1;        // ... and so have everybody else's
    for (auto& kv : st->opened) (void)hipIpcCloseMemHandle(kv.second);
    st->opened.clear();
    st->exported.clear();
    st->send_chan.clear();
    st->recv_chan.clear();
    st->next_chan.assign(st->nranks, 0);
    if (hipMemset(st->mailbox, 0, st->mailbox_words * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 1;
    z = 0;
    if (tcp_allreduce(st->mesh, 0, &z) != 0) return 1;        // nobody raises a flag in a mailbox that is still to be zeroed
    st->stale = false;
    return 0;
}

int ipc_start(void* user, int n, const ykh::HaloMsg* m, void* stream_) {
    IpcState* st = static_cast<IpcState*>(user);
    hipStream_t stream = (hipStream_t)stream_;
    st->wait_ptr.clear(); st->wait_val.clear();
    for (int i = 0; i < n; i++)
        if (m[i].tag < 0 || m[i].tag >= TAGS || m[i].peer < 0 || m[i].peer >= st->nranks) {
            fprintf(stderr, "yask ipc transport: message %d (peer %d, tag %d) is outside the mailbox layout\n", i, m[i].peer, m[i].tag);
            return 1;
        }
    // ---- host side, FIRST USE of a channel only: tell the sender where its data goes (all sends first, then the receives:
    // 96-byte messages, far below a socket buffer)
    std::vector<RecvChan*> rc(n, nullptr);
    std::map<int, bool> sent_to;                              // peers that got a registration from me in THIS exchange
    for (int i = 0; i < n; i++) {
        if (m[i].peer == st->rank || !m[i].recv_bytes) continue;
        const long long k = chan_key(m[i].peer, 26 - m[i].tag, m[i].key);       // the channel is named by the tag its SENDER uses
        auto it = st->recv_chan.find(k);
        if (it == st->recv_chan.end()) {
            Registration rg{};
            rg.tag = 26 - m[i].tag; rg.key = m[i].key; rg.chan = st->next_chan[m[i].peer];
            if (rg.chan >= CHANS) { fprintf(stderr, "yask ipc transport: rank %d needs more than %d channels from rank %d\n", st->rank, CHANS, m[i].peer); return 1; }
            if (!export_buf(st, m[i].recv_buf, &rg.handle, &rg.offset, &rg.room)) { fprintf(stderr, "yask ipc transport: cannot export a receive buffer\n"); return 1; }
            if (rg.room < m[i].recv_bytes) { fprintf(stderr, "yask ipc transport: a receive buffer is shorter than its message\n"); return 1; }
            if (!send_all(st->mesh->fd[m[i].peer], &rg, sizeof(rg))) return 1;
            if (st->verbose > 1) fprintf(stderr, "ipc[%d]: registered (sender's tag %d, key %d) as channel %d with rank %d, %zu bytes\n", st->rank, rg.tag, rg.key, rg.chan, m[i].peer, m[i].recv_bytes);
            st->ctl_msgs++; st->ctl_bytes += (long long)sizeof(rg);
            sent_to[m[i].peer] = true;
            st->next_chan[m[i].peer]++;
            it = st->recv_chan.emplace(k, RecvChan{rg.chan, m[i].recv_buf, 0u}).first;
        } else if (it->second.buf != m[i].recv_buf) {
            fprintf(stderr, "yask ipc transport: rank %d: the receive buffer of (peer %d, tag %d, key %d) moved without a reset of the transport\n",
                    st->rank, m[i].peer, m[i].tag, m[i].key);
            return 1;
        }
        rc[i] = &it->second;
    }
    // The other direction: addresses of the peers' buffers for MY messages.  Mapping a peer's allocation (hipIpcOpenMemHandle) is
    // ORDERED between the two ends of a link: when both map each other's buffers in the same exchange, the lower rank maps first
    // and then sends a token, the higher rank waits for it.  Four bench.py ranks, released from a barrier at the same instant,
    // each sat in hipIpcOpenMemHandle on its x neighbour's 2.6 GB var while that neighbour sat in the same call on theirs --
    // for good (gpurun_out/r4t: 0 <-> 1, 2 <-> 3); ranks that arrive a few microseconds apart never showed it.  A rank only ever
    // waits for LOWER ranks, so the waits cannot form a cycle; all of this happens at the first use of a channel only.
    std::vector<SendChan*> sc(n, nullptr);
    std::map<int, std::vector<Registration>> pending;      // peer -> registrations read, not yet mapped
    std::map<int, bool> token_from;
    auto read_record = [&](int peer) -> bool {                // one record from a peer: a registration (kept) or a token
        Registration rg{};
        if (!recv_all(st->mesh->fd[peer], &rg, sizeof(rg))) { fprintf(stderr, "yask ipc transport: rank %d lost the control link to rank %d\n", st->rank, peer); return false; }
        if (rg.tag == -1) { token_from[peer] = true; return true; }
        if (rg.tag < 0 || rg.tag >= TAGS || rg.chan < 0 || rg.chan >= CHANS) { fprintf(stderr, "yask ipc transport: malformed registration from rank %d\n", peer); return false; }
        pending[peer].push_back(rg);
        return true;
    };
    for (int i = 0; i < n; i++) {
        if (m[i].peer == st->rank || !m[i].send_bytes) continue;
        const int peer = m[i].peer;
        if (st->send_chan.count(chan_key(peer, m[i].tag, m[i].key))) continue;
        // the receiver posts the same exchange: its registration for this channel is on its way (others of the link may come first)
        auto have = [&]() { for (const Registration& r : pending[peer]) if (r.tag == m[i].tag && r.key == m[i].key) return true; return false; };
        while (!have()) {
            if (st->verbose > 1) fprintf(stderr, "ipc[%d]: waiting for rank %d's registration of (tag %d, key %d), %zu bytes to send\n", st->rank, peer, m[i].tag, m[i].key, m[i].send_bytes);
            if (!read_record(peer)) return 1;
        }
    }
    for (auto& kv : pending) {                                 // (std::map: peers in increasing rank order)
        const int peer = kv.first;
        const bool mutual = sent_to.count(peer) != 0;          // the peer maps MY buffers in this exchange as well
        if (mutual && peer < st->rank)
            while (!token_from[peer]) {
                if (st->verbose > 1) fprintf(stderr, "ipc[%d]: waiting for rank %d to finish mapping my buffers\n", st->rank, peer);
                if (!read_record(peer)) return 1;
            }
        for (const Registration& rg : kv.second) {
            if (st->verbose > 1) fprintf(stderr, "ipc[%d]: mapping rank %d's buffer of (tag %d, key %d)\n", st->rank, peer, rg.tag, rg.key);
            void* base = import_buf(st, peer, rg.handle);
            if (!base) return 1;
            st->send_chan[chan_key(peer, rg.tag, rg.key)] = SendChan{rg.chan, (char*)base + rg.offset, rg.room, 0u};
        }
        if (mutual && peer > st->rank) {
            Registration tok{};
            tok.tag = -1;
            if (!send_all(st->mesh->fd[peer], &tok, sizeof(tok))) return 1;
            st->ctl_msgs++; st->ctl_bytes += (long long)sizeof(tok);
        }
    }
    for (int i = 0; i < n; i++) {
        if (m[i].peer == st->rank || !m[i].send_bytes) continue;
        auto it = st->send_chan.find(chan_key(m[i].peer, m[i].tag, m[i].key));
        if (it == st->send_chan.end()) { fprintf(stderr, "yask ipc transport: rank %d has no address for (peer %d, tag %d, key %d)\n", st->rank, m[i].peer, m[i].tag, m[i].key); return 1; }
        if (it->second.room < m[i].send_bytes) {
            fprintf(stderr, "yask ipc transport: rank %d: message of %zu bytes for (peer %d, tag %d, key %d) exceeds the %llu bytes registered\n",
                    st->rank, m[i].send_bytes, m[i].peer, m[i].tag, m[i].key, it->second.room);
            return 1;
        }
        sc[i] = &it->second;
    }
    if (st->verbose > 1) fprintf(stderr, "ipc[%d]: exchange of %d messages: addresses known, enqueueing\n", st->rank, n);
    // ---- device side, on the comm stream
    std::vector<unsigned*> sp;
    std::vector<const unsigned*> wp;
    std::vector<unsigned> sv, wv;
    // (1) my receive buffers may be written: clear-to-send to every sender (its cts word lives in ITS mailbox, slot [me])
    for (int i = 0; i < n; i++) {
        if (!rc[i]) continue;
        const unsigned e = ++rc[i]->epoch;
        sp.push_back(st->word(st->peer_mailbox[m[i].peer], st->rank, rc[i]->chan, 0));
        sv.push_back(e);
        st->wait_ptr.push_back(st->word(st->mailbox, m[i].peer, rc[i]->chan, 1));     // its `ready` word: mine, slot [sender]
        st->wait_val.push_back(e);
    }
    if (!sp.empty()) { ykh::launch_set_words((int)sp.size(), sp.data(), sv.data(), stream); st->dev_ops++; }
    // (2) wait until the receivers of MY messages are clear, copy, raise their `ready` words
    sp.clear(); sv.clear();
    for (int i = 0; i < n; i++) {
        if (!sc[i]) continue;
        wp.push_back(st->word(st->mailbox, m[i].peer, sc[i]->chan, 0));
        wv.push_back(++sc[i]->epoch);
    }
    if (!wp.empty()) { ykh::launch_wait_words((int)wp.size(), wp.data(), wv.data(), st->err(), st->timeout_s, stream); st->dev_ops++; }
    for (int i = 0; i < n; i++) {
        if (m[i].peer == st->rank) {            // loop-back (yk_env_transport_loopback): a plain device-to-device copy
            if (m[i].send_bytes != m[i].recv_bytes) return 1;
            if (m[i].send_bytes && hipMemcpyAsync(m[i].recv_buf, m[i].send_buf, m[i].send_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
            continue;
        }
        if (!sc[i]) continue;
        // (hipMemcpyDefault: the destination is another process's allocation, possibly on another device -- the runtime picks the
        //  path, SDMA over xGMI or a blit, from the pointers)
        if (hipMemcpyAsync(sc[i]->dst, m[i].send_buf, m[i].send_bytes, hipMemcpyDefault, stream) != hipSuccess) {
            fprintf(stderr, "yask ipc transport: copy into rank %d's buffer failed: %s\n", m[i].peer, hipGetErrorString(hipGetLastError()));
            return 1;
        }
        st->dev_ops++;
        sp.push_back(st->word(st->peer_mailbox[m[i].peer], st->rank, sc[i]->chan, 1));
        sv.push_back(sc[i]->epoch);
    }
    if (!sp.empty()) { ykh::launch_set_words((int)sp.size(), sp.data(), sv.data(), stream); st->dev_ops++; }
    return 0;
}
// (3) the stream goes on (unpack kernels) when every message of the exchange has landed
int ipc_wait(void* user, int, const ykh::HaloMsg*, void* stream) {
    IpcState* st = static_cast<IpcState*>(user);
    if (!st->wait_ptr.empty()) {
        ykh::launch_wait_words((int)st->wait_ptr.size(), st->wait_ptr.data(), st->wait_val.data(), st->err(), st->timeout_s, (hipStream_t)stream);
        st->dev_ops++;
    }
    st->wait_ptr.clear(); st->wait_val.clear();
    return 0;
}
// a waiter gave up: say which flags never came (the streams have drained when this runs)
void dump_mailbox(IpcState* st) {
    std::vector<unsigned> w(st->mailbox_words, 0);
    if (hipMemcpy(w.data(), st->mailbox, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    fprintf(stderr, "yask ipc transport: rank %d mailbox (%s memory, job on %s): channel / epoch expected / word found\n", st->rank,
            KIND_NAME[st->kind], st->multi_device ? "several devices" : "one device");
    for (auto& kv : st->send_chan) {
        const int peer = (int)((kv.first >> 32) / TAGS), tag = (int)((kv.first >> 32) % TAGS), key = (int)(unsigned)kv.first;
        const unsigned got = w[((size_t)peer * CHANS + kv.second.chan) * 2 + 0];
        fprintf(stderr, "   send to rank %d tag %2d key %3d chan %3d: clear-to-send expected >= %u, found %u%s\n", peer, tag, key, kv.second.chan, kv.second.epoch, got,
                (int)(got - kv.second.epoch) < 0 ? "   <-- never arrived" : "");
    }
    for (auto& kv : st->recv_chan) {
        const int peer = (int)((kv.first >> 32) / TAGS), tag = (int)((kv.first >> 32) % TAGS), key = (int)(unsigned)kv.first;
        const unsigned got = w[((size_t)peer * CHANS + kv.second.chan) * 2 + 1];
        fprintf(stderr, "   recv from rank %d tag %2d key %3d chan %3d: ready expected >= %u, found %u%s\n", peer, tag, key, kv.second.chan, kv.second.epoch, got,
                (int)(got - kv.second.epoch) < 0 ? "   <-- never arrived" : "");
    }
}
int ipc_check(void* user) {
    IpcState* st = static_cast<IpcState*>(user);
    unsigned e = 0;
    if (hipMemcpy(&e, st->err(), sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (e) {
        (void)hipMemset(st->err(), 0, sizeof(unsigned));
        fprintf(stderr, "yask ipc transport: rank %d waited in vain for a neighbour's flag (%.0f s)\n", st->rank, st->timeout_s);
        dump_mailbox(st);
        st->stale = true;        // (the epochs of this rank and its peers no longer agree: start over at the next collective point)
        return 1;
    }
    return 0;
}
int ipc_allreduce(void* user, int op, long long* val) { return tcp_allreduce(static_cast<IpcState*>(user)->mesh, op, val); }
int ipc_counters(void* user, long long* out, int cap) {
    IpcState* st = static_cast<IpcState*>(user);
    const long long v[6] = {st->ctl_msgs, st->ctl_bytes, st->begins, st->resets, st->dev_ops, (long long)st->kind};
    for (int i = 0; i < cap && i < 6; i++) out[i] = v[i];
    return 6;
}

void unmap_mailbox(int kind, unsigned* dev, void* host, size_t bytes, bool mine) {
    if (kind == 3) {
        if (host) { (void)hipHostUnregister(host); (void)munmap(host, bytes); }
    } else if (dev) {
        if (mine) (void)hipFree(dev); else (void)hipIpcCloseMemHandle(dev);
    }
}
void ipc_free(void* p) {
    IpcState* st = static_cast<IpcState*>(p);
    if (ipc_wedged.load()) {
        // a helper thread is stuck inside hipIpcOpenMemHandle (open_guarded): no call that may queue behind it.  The host-side
        // resources go; the device-side ones (mappings, the mailbox) stay until the process ends.
        if (!st->shm_name.empty()) (void)shm_unlink(st->shm_name.c_str());
        if (st->mesh) { for (int fd : st->mesh->fd) if (fd >= 0) ::close(fd); delete st->mesh; }
        delete st;
        return;
    }
    (void)hipDeviceSynchronize();
    const size_t mb = st->mailbox_words * sizeof(unsigned);
    for (auto& kv : st->opened) (void)hipIpcCloseMemHandle(kv.second);
    for (int r = 0; r < (int)st->peer_mailbox.size(); r++)
        if (r != st->rank && st->peer_mailbox[r]) unmap_mailbox(st->peer_kind[r], st->peer_mailbox[r], st->peer_host[r], mb, false);
    // NOT a collective (round 4): an env is released when its last holder goes, which a garbage-collected host (Python) decides
    // per rank and at any time -- a barrier here deadlocked bench.py (two ranks in this destructor, two in a torch.distributed
    // barrier; gpurun_out/r4f).  Nothing needs one: a peer's mapping of my mailbox keeps the memory behind it alive after my
    // hipFree (IPC mappings hold a reference on the allocation), so flags it still stores land in memory nobody reads.
    if (st->mailbox) unmap_mailbox(st->kind, st->mailbox, st->mailbox_host, mb, true);
    if (!st->shm_name.empty()) (void)shm_unlink(st->shm_name.c_str());
    if (st->mesh) { for (int fd : st->mesh->fd) if (fd >= 0) ::close(fd); delete st->mesh; }
    delete st;
}

// pinned host memory behind a shm segment: created (mine) or opened (a peer's), registered with this rank's device
unsigned* map_host_mailbox(const char* name, size_t bytes, bool create, void** host_out) {
    const int fd = shm_open(name, create ? (O_CREAT | O_TRUNC | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return nullptr;
    // (a segment this call created and cannot use must not stay behind in /dev/shm)
    if (create && ftruncate(fd, (off_t)bytes) != 0) { ::close(fd); (void)shm_unlink(name); return nullptr; }
    void* h = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (h == MAP_FAILED) { if (create) (void)shm_unlink(name); return nullptr; }
    if (create) std::memset(h, 0, bytes);
    void* d = nullptr;
    if (hipHostRegister(h, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        (void)munmap(h, bytes);
        if (create) (void)shm_unlink(name);
        return nullptr;
    }
    *host_out = h;
    return (unsigned*)d;
}

}  // namespace

extern "C" {

int yk_env_init_ipc(yk_env_h e, int rank, int nranks, const char* addr, int base_port) {
    try {
        if (!e) return 1;
        e->env->set_ranks(rank, nranks);
        // every failing return below releases what was built so far (mesh sockets, mailbox, the shm segment's NAME, peers' mailboxes
        // already mapped): the state is owned by this guard until the transport is installed
        struct Guard { IpcState* st; ~Guard() { if (st) ipc_free(st); } } guard{new IpcState};
        IpcState* st = guard.st;
        st->rank = rank; st->nranks = nranks;
        if (const char* t = getenv("YASK_HIP_WAIT_TIMEOUT_S")) { const double v = atof(t); if (v > 0) st->timeout_s = v; }
        if (const char* t = getenv("YASK_HIP_IPC_VERBOSE")) st->verbose = atoi(t);
        st->mesh = tcp_connect_mesh(rank, nranks, addr && *addr ? addr : "127.0.0.1", base_port);
        if (!st->mesh) { fprintf(stderr, "yask ipc transport: rank %d could not connect the control mesh\n", rank); return 1; }
        st->next_chan.assign(nranks, 0);
        // ---- which device does every rank sit on?  (bus ids through rank 0's all-reduce-free mesh: pairwise, lower rank first)
        MailboxAd mine{};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetPCIBusId(mine.busid, (int)sizeof(mine.busid) - 1, dev) != hipSuccess) { (void)hipGetLastError(); snprintf(mine.busid, sizeof(mine.busid), "device-%d", dev); }
        std::vector<MailboxAd> theirs(nranks);
        auto swap_ads = [&]() -> bool {
            for (int p = 0; p < nranks; p++) {
                if (p == rank) continue;
                const int fd = st->mesh->fd[p];
                const bool ok = rank < p ? (send_all(fd, &mine, sizeof(mine)) && recv_all(fd, &theirs[p], sizeof(mine)))
                                         : (recv_all(fd, &theirs[p], sizeof(mine)) && send_all(fd, &mine, sizeof(mine)));
                if (!ok) { fprintf(stderr, "yask ipc transport: handle exchange with rank %d failed\n", p); return false; }
            }
            return true;
        };
        if (!swap_ads()) return 1;
        for (int p = 0; p < nranks; p++)
            if (p != rank && std::strncmp(theirs[p].busid, mine.busid, sizeof(mine.busid)) != 0) st->multi_device = true;
        // ---- my mailbox: the first kind of memory that can be allocated AND shared wins; plain device memory never across devices
        st->mailbox_words = (size_t)nranks * WORDS_PER_PEER + 1;
        const size_t mb = st->mailbox_words * sizeof(unsigned);
        const char* want = getenv("YASK_HIP_MAILBOX");
        int only = -1;
        if (want && *want) {
            only = !std::strcmp(want, "uncached") ? 0 : !std::strcmp(want, "finegrained") ? 1 : !std::strcmp(want, "plain") ? 2 : !std::strcmp(want, "host") ? 3 : -2;
            if (only == -2) { fprintf(stderr, "yask ipc transport: YASK_HIP_MAILBOX=%s is not one of uncached, finegrained, plain, host\n", want); return 1; }
        }
        if (only == 2 && st->multi_device) {
            fprintf(stderr, "yask ipc transport: rank %d: a mailbox in plain (cached) device memory is refused when ranks sit on different devices\n", rank);
            return 1;
        }
        for (int k = 0; k < 4 && st->kind < 0; k++) {
            if (only >= 0 && k != only) continue;
            if (k == 2 && st->multi_device) continue;        // flags other devices poll must not sit in this device's L2
            if (k == 3) {
                st->shm_name = "/yask_mbx_" + std::to_string(base_port) + "_" + std::to_string((long)getpid()) + "_" + std::to_string(rank);
                unsigned* d = map_host_mailbox(st->shm_name.c_str(), mb, true, &st->mailbox_host);
                if (!d) { st->shm_name.clear(); continue; }
                st->mailbox = d;
                snprintf(mine.shm, sizeof(mine.shm), "%s", st->shm_name.c_str());
                st->kind = 3;
                break;
            }
            void* p = nullptr;
            const hipError_t rc = k == 2 ? hipMalloc(&p, mb) : hipExtMallocWithFlags(&p, mb, k == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            if (rc != hipSuccess || !p) { (void)hipGetLastError(); continue; }
            if (hipMemset(p, 0, mb) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&mine.handle, p) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipFree(p);
                continue;
            }
            st->mailbox = (unsigned*)p;
            st->kind = k;
        }
        if (st->kind < 0) {
            fprintf(stderr, "yask ipc transport: rank %d cannot allocate and share a mailbox%s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)\n", rank,
                    st->multi_device ? " that other devices see coherently" : "");
            return 1;
        }
        mine.kind = st->kind;
        // ---- every rank maps every other rank's mailbox
        if (!swap_ads()) return 1;
        st->peer_mailbox.assign(nranks, nullptr);
        st->peer_host.assign(nranks, nullptr);
        st->peer_kind.assign(nranks, -1);
        st->peer_mailbox[rank] = st->mailbox;
        st->peer_kind[rank] = st->kind;
        for (int p = 0; p < nranks; p++) {
            if (p == rank) continue;
            st->peer_kind[p] = theirs[p].kind;
            if (theirs[p].kind == 3) {
                theirs[p].shm[sizeof(theirs[p].shm) - 1] = 0;
                st->peer_mailbox[p] = map_host_mailbox(theirs[p].shm, mb, false, &st->peer_host[p]);
                if (!st->peer_mailbox[p]) { fprintf(stderr, "yask ipc transport: rank %d cannot map the host mailbox of rank %d\n", rank, p); return 1; }
                continue;
            }
            if (theirs[p].kind == 2 && std::strncmp(theirs[p].busid, mine.busid, sizeof(mine.busid)) != 0) {
                fprintf(stderr, "yask ipc transport: rank %d refuses the plain-memory mailbox of rank %d on another device\n", rank, p);
                return 1;
            }
            void* base = nullptr;
            const hipError_t orc = open_guarded(&base, theirs[p].handle, st->open_limit_s());
            if (orc != hipSuccess) {
                fprintf(stderr, "yask ipc transport: rank %d cannot map the mailbox of rank %d: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)\n", rank, p, open_error(orc));
                return 1;
            }
            st->peer_mailbox[p] = (unsigned*)base;
        }
        // (everybody has mapped everybody: the shm names can go)
        { long long z = 0; if (tcp_allreduce(st->mesh, 0, &z) != 0) return 1; }
        if (!st->shm_name.empty()) { (void)shm_unlink(st->shm_name.c_str()); st->shm_name.clear(); }
        if (e->env->trace || getenv("YASK_HIP_IPC_VERBOSE"))
            fprintf(stderr, "yask ipc transport: rank %d of %d up on device %s (%s), mailbox in %s memory\n", rank, nranks, mine.busid,
                    st->multi_device ? "job spans devices" : "all ranks on this device", KIND_NAME[st->kind]);
        // (no peer maps var storage unless asked to: YASK_HIP_IPC_DIRECT=1 brings the in-place x faces back, see Env::direct_halo_ok)
        // -- and only when EVERY rank asks: a rank that packs facing one that maps would exchange different messages
        long long direct = 0;
        { const char* d = getenv("YASK_HIP_IPC_DIRECT"); direct = d && atoi(d) != 0; }
        if (tcp_allreduce(st->mesh, 1, &direct) != 0) return 1;       // op 1 = min
        e->env->drop_transport();
        e->env->exch_start = ipc_start;
        e->env->exch_wait = ipc_wait;
        e->env->exch_reset = ipc_reset;
        e->env->exch_begin = ipc_begin;
        e->env->exch_check = ipc_check;
        e->env->exch_counters = ipc_counters;
        e->env->direct_halo_ok = direct != 0;
        e->env->allreduce = ipc_allreduce;
        e->env->user = st;
        e->env->user_free = ipc_free;
        guard.st = nullptr;
        return 0;
    } catch (...) { return 1; }
}

}  // extern "C"
