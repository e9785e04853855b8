// mgs_torch.cpp -- the COMPILED binding of the rasterizer's autograd path: manigaussian_amd/_mgs_torch.so.
//
// What the reference's own binding is (RAST/rasterize_points.cu:35-128 RasterizeGaussiansCUDA, :130-225
// RasterizeGaussiansBackwardCUDA; RAST = third_party/gaussian-splatting/submodules/diff-gaussian-rasterization): compiled code
// that takes torch tensors, allocates outputs and workspaces, and calls the rasterizer.  Rounds 1-4 did that job in Python over
// ctypes (manigaussian_amd/_C.py) at 125-133 us of host time per forward + backward -- as much as the 150 us of GPU work, so
// ManiGaussian's eager caller (MG/neural_rendering.py:283,324 -> MG/gaussian_renderer/__init__.py:74) was host-bound.  This file
// is the same job in C++ over the SAME C ABI (include/mgsplat.h; no kernel, no HIP code here): argument marshalling, ONE
// workspace arena, the pinned status slots, the workspace marks, and a torch::autograd::Node whose backward the autograd engine
// calls without entering Python.  _C.py stays: it is the raw three-function surface of the reference's `_C` module, the
// multi-view path, HIP-graph capture, debug / prefiltered calls and the fallback when this module is not built.
//
// Host-side protocol (mirrors manigaussian_amd/_state.py, which documents it at length):
//   mode safe   shapes whose worst-case workspace fits the budget: enqueue and return (cannot overflow).  Every other shape:
//               workspace from the shape's marks, everything enqueued, then wait for the PREPROCESS's report only (status
//               words 0 and 2); too small -> bin + render again with room (what RAST/cuda_rasterizer/rasterizer_impl.cu:282-289
//               does by blocking on a cudaMemcpy -- here binning and render are already running while the call returns).
//   mode async  workspace from the marks x head-room, nothing waited for; overflow repaired at backward entry or raised late.
//   mode blocking, debug, prefiltered, graph capture, padded feature widths, non-contiguous inputs: not handled here (None).
#include <Python.h>
#include <torch/extension.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/saved_variable.h>
#include <torch/csrc/autograd/variable.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mgsplat.h"

namespace py = pybind11;
using torch::autograd::Node;
using torch::autograd::SavedVariable;
using torch::autograd::variable_list;

namespace {

constexpr int NSLOTS = 1024;         // status slots per device (four 64-bit words each: three used)
constexpr int64_t SLOT_WORDS = 4;
enum Mode { MODE_SAFE = 0, MODE_ASYNC = 1, MODE_BLOCKING = 2 };
enum Policy { POLICY_REPAIR = 0, POLICY_RAISE = 1 };

struct Config {
  std::atomic<bool> enabled{true};
  std::atomic<int> mode{MODE_SAFE};
  std::atomic<int> policy{POLICY_REPAIR};
  double head_inst = 1.5, head_chunks = 2.0;
  int64_t safe_bytes = -1;  // < 0: per device, 1/32 of its memory and at least 1 GB (manigaussian_amd/_state.py safe_bytes)
  MgsOptions opt;       // per-call options every forward carries (manigaussian_amd._lib.DEFAULT_OPTIONS)
  std::mutex mu;
};
Config& cfg() { static Config c; return c; }

struct Counters {
  std::atomic<int64_t> forwards{0}, backwards{0}, declined{0}, waited{0}, retried{0}, recovered{0}, budget_fallbacks{0};
};
Counters& counters() { static Counters c; return c; }

// shape key of the workspace marks: V = 0 single view (this file), V >= 1 a batch of V views (manigaussian_amd/views.py)
struct Key {
  int32_t V, P, W, H, F, tight;
  bool operator==(const Key& o) const { return V == o.V && P == o.P && W == o.W && H == o.H && F == o.F && tight == o.tight; }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (int32_t v : {k.V, k.P, k.W, k.H, k.F, k.tight}) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001b3ull + 0x632be59bd9b4e019ull;
    return (size_t)h;
  }
};
struct Mark { int64_t R = 0; int64_t chunks = -1; };  // chunks < 0: unknown (worst-case pool)

// One forward whose device report has not been folded into the marks yet.
struct PendingRec {
  MgsRasterArgs a;           // what THIS run was given (status_tag, binning_capacity, chunk_pool, binning_bytes, shape)
  uint64_t* slot = nullptr;  // three pinned words
  Key key{};
  int32_t num_rendered = -1, chunks_used = -1, ref_rendered = -1;
  int rc = MGS_PENDING;
  bool recoverable = false, recovered = false, backward_enqueued = false;
};

// ---- where the host time of a call goes (diagnostics: set_profile(true), profile_read()) ----
enum Seg { SG_CHECKS = 0, SG_STREAM, SG_DRAIN, SG_SIZES, SG_ALLOC, SG_ARGS, SG_LIBRARY, SG_NODE, SG_RETURN,
           SB_UNPACK, SB_SETTLE, SB_ALLOC, SB_LIBRARY, SB_VIEWS, SG_COUNT };
const char* const kSegNames[SG_COUNT] = {"fwd.checks", "fwd.guard+stream+capture", "fwd.drain", "fwd.sizes", "fwd.alloc",
                                         "fwd.args", "fwd.library", "fwd.node", "fwd.return",
                                         "bwd.unpack+cotangents", "bwd.settle", "bwd.alloc", "bwd.library", "bwd.views"};
std::atomic<bool> g_profile{false};
std::atomic<int64_t> g_seg_ns[SG_COUNT];
std::atomic<int64_t> g_seg_n[SG_COUNT];
struct SegClock {
  bool on;
  std::chrono::steady_clock::time_point t;
  SegClock() : on(g_profile.load(std::memory_order_relaxed)) { if (on) t = std::chrono::steady_clock::now(); }
  void lap(int seg) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    g_seg_ns[seg] += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count();
    g_seg_n[seg]++;
    t = n;
  }
};

constexpr int MAX_DEVICES = 64;
std::atomic<int> g_py_pending[MAX_DEVICES];  // per device: forwards the Python shim has registered and not read yet (cross-drain hint)
py::object* g_py_drain = nullptr;  // manigaussian_amd._state: drains the Python side's reports of a device

const char* kOutgrew = "an asynchronous rasterizer forward outgrew the workspace sized from earlier calls of the same shape";
const char* kHandshake = "an asynchronous rasterizer forward's preprocess gave up waiting for its zeroed tile tables (a workgroup of "
                         "the launch made no progress for about a second) and binned nothing";

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error(what); }

void check_rc(int rc, const char* what) {
  if (rc != MGS_OK) fail(std::string(what) + ": " + mgs_last_error() + " (code " + std::to_string(rc) + ")");
}

// Release the GIL for a blocking wait -- only if this thread holds it (the autograd engine's threads do not).
struct MaybeReleaseGil {
  PyThreadState* saved = nullptr;
  MaybeReleaseGil() { if (PyGILState_Check()) saved = PyEval_SaveThread(); }
  ~MaybeReleaseGil() { if (saved) PyEval_RestoreThread(saved); }
};

void warn_runtime(const std::string& msg) {
  py::gil_scoped_acquire gil;
  if (PyErr_WarnEx(PyExc_RuntimeWarning, msg.c_str(), 2) < 0) throw py::error_already_set();
}

struct DeviceState {
  int index;
  at::Tensor status;  // pinned int64 [NSLOTS * SLOT_WORDS]
  uint64_t* base = nullptr;
  int next_slot = 0;
  uint32_t tag = 0x4000;  // (the Python ring counts from 0: tags of the two rings rarely meet -- and never share a slot)
  std::unordered_map<Key, Mark, KeyHash> marks;
  std::deque<std::shared_ptr<PendingRec>> pending;
  std::mutex mu;
  bool python_side_built = false;  // manigaussian_amd._state.DeviceState of this device exists (see rasterize())
  int64_t auto_safe_bytes = 0;     // the default worst-case-workspace budget of this device
  // bytes of WORST-CASE workspaces that live forwards still hold on this device (a node keeps its workspace until its backward
  // has run or the graph is freed; the ctypes shim adds its own through hold_add): the `safe` budget is charged against this
  // sum, not per call -- V forwards before the first backward would otherwise pin V x 4 GB at BASELINE configs[2]
  std::atomic<int64_t> held_bytes{0};

  explicit DeviceState(int idx) : index(idx) {}

  void ensure_ring() {
    if (base) return;
    status = at::full({NSLOTS * SLOT_WORDS}, -1, at::TensorOptions().dtype(at::kLong).device(at::kCPU)).pin_memory();
    base = reinterpret_cast<uint64_t*>(status.data_ptr<int64_t>());
  }
  uint64_t* take_slot(uint32_t* tag_out) {  // mu held
    ensure_ring();
    uint64_t* p = base + (size_t)next_slot * SLOT_WORDS;
    next_slot = (next_slot + 1) % NSLOTS;
    tag = (tag + 1) & 0xffffu;
    *tag_out = tag;
    return p;
  }
  // ---- marks (mu held) ----
  void learn(const Key& k, int64_t R, int64_t chunks, bool pool_unknown) {
    Mark& m = marks[k];
    if (R >= 0 && R > m.R) m.R = R;
    if (pool_unknown) m.chunks = -1;
    else if (chunks >= 0 && chunks > m.chunks) m.chunks = chunks;
  }
  bool guess(const Key& k, int64_t* cap, int64_t* pool) {
    auto it = marks.find(k);
    if (it == marks.end() || it->second.chunks < 0) return false;
    *cap = (int64_t)(it->second.R * cfg().head_inst) + 4096;
    *pool = (int64_t)(it->second.chunks * cfg().head_chunks) + 64;
    return true;
  }
};

// A worst-case workspace charged against its device's budget for as long as somebody holds this object.
struct Lease {
  DeviceState* st;
  int64_t bytes;
  Lease(DeviceState* s, int64_t b) : st(s), bytes(b) { st->held_bytes += b; }
  ~Lease() { st->held_bytes -= bytes; }
  Lease(const Lease&) = delete;
  Lease& operator=(const Lease&) = delete;
};

std::mutex g_states_mu;
std::vector<std::unique_ptr<DeviceState>> g_states;
DeviceState& state(int idx) {
  std::lock_guard<std::mutex> lk(g_states_mu);
  if ((int)g_states.size() <= idx) g_states.resize(idx + 1);
  if (!g_states[idx]) g_states[idx] = std::make_unique<DeviceState>(idx);
  return *g_states[idx];
}

// non-blocking read of a forward's status words (manigaussian_amd/_state.py Pending.poll)
int poll(PendingRec& p) {
  if (p.rc != MGS_PENDING || p.recovered) return p.rc;
  const volatile uint64_t* w = p.slot;
  if (w[0] == ~0ull && w[1] == ~0ull) return p.rc;
  int32_t nr = -1, ch = -1, ref = -1;
  const int rc = mgs_forward_result(&p.a, p.slot, &nr, &ch, &ref);
  if (nr >= 0) p.num_rendered = nr;
  if (ch >= 0) p.chunks_used = ch;
  if (ref >= 0) p.ref_rendered = ref;
  if (rc != MGS_PENDING) p.rc = rc;
  return rc;
}

// Fold a finished forward into the marks (mu held); returns an error message if the call must raise ("" otherwise) and
// appends warnings to `warnings` (issued by the caller outside the lock).
std::string account(DeviceState& st, PendingRec& p, int rc, std::vector<std::string>* warnings) {
  if (rc == MGS_OK) { st.learn(p.key, p.num_rendered, p.chunks_used, false); return ""; }
  if (rc == MGS_NEED_CAPACITY || rc == MGS_RETRY_TABLE_INIT) {
    std::string msg;
    if (rc == MGS_NEED_CAPACITY) {
      const int cap = p.a.binning_capacity;
      const bool over_inst = p.num_rendered > cap && cap > 0;
      st.learn(p.key, p.num_rendered, -1, !over_inst);
      std::string what = over_inst ? std::to_string(p.num_rendered) + " (Gaussian, tile) instances > capacity " + std::to_string(cap)
                                   : "chunk pool of " + std::to_string(p.a.chunk_pool) + " records";
      msg = std::string(kOutgrew) + " (" + what + "): the images of THAT call were incomplete.  The marks are raised";
    } else {  // nothing to learn: the scene did not grow, a workgroup of the preprocess launch was held back
      msg = std::string(kHandshake) + ": the images of THAT call were incomplete";
    }
    const int policy = cfg().policy.load();
    if (p.recovered || (policy == POLICY_REPAIR && p.recoverable && !p.backward_enqueued)) {
      warnings->push_back(msg + "; the call's backward re-renders on the blocking path before it runs, but a loss computed "
                                "from those images was computed from incomplete images." +
                          (rc == MGS_NEED_CAPACITY ? "  For scenes that grow abruptly use manigaussian_amd.set_forward_mode('safe') "
                                                     "(the default) or a larger set_headroom()."
                                                   : "  manigaussian_amd.set_options(table_init=1) removes the hand-shake."));
      return "";
    }
    if (p.recoverable && !p.backward_enqueued) {  // overflow policy "raise"
      p.recovered = true;                         // its backward, if it still comes, raises too
      return msg + "; the step is lost (overflow policy 'raise'): re-run it" +
             (rc == MGS_NEED_CAPACITY ? ", or use manigaussian_amd.set_forward_mode('safe') (the default) for scenes that grow abruptly."
                                      : ".");
    }
    return msg + " and its gradients were computed on the incomplete state; re-run the step" +
           (rc == MGS_NEED_CAPACITY ? ", or use manigaussian_amd.set_forward_mode('safe') (the default) for scenes that grow abruptly."
                                    : " (manigaussian_amd.set_options(table_init=1) removes the hand-shake).");
  }
  return std::string("rasterizer forward failed: ") + mgs_last_error() + " (code " + std::to_string(rc) + ")";
}

// Read every report that has arrived (wait: all of them, synchronising with the device once if one is outstanding).
// Three steps, so that the device's mutex is NEVER held across the GIL's release / re-acquisition -- the lock order is GIL ->
// mutex everywhere (a second Python thread on the same device blocks on the mutex while it holds the GIL): poll under the lock;
// synchronise outside it; account under the lock again.  `deferred`: append the warnings there instead of issuing them (callers
// that hold a lock of their own -- a node's -- issue them after dropping it: warn_runtime takes the GIL).
void drain(DeviceState& st, bool wait, std::vector<std::string>* deferred = nullptr) {
  std::string failed;
  std::vector<std::string> warnings;
  // reports that were outstanding when the device was synchronised.  Shared pointers, not addresses: another thread's drain may
  // account and free a record in the meantime, and a forward enqueued AFTER the synchronisation may be allocated at that very
  // address -- it would pass for "must have reported" (found by the two-thread stress test)
  std::vector<std::shared_ptr<PendingRec>> must_finish;
  bool need_sync = false;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    if (st.pending.empty()) return;
    const size_t n = st.pending.size();
    size_t kept = 0;
    for (size_t i = 0; i < n; i++) {
      const std::shared_ptr<PendingRec>& p = st.pending[i];
      if (poll(*p) != MGS_PENDING) continue;
      if (wait || (n - i - 1) + kept >= (size_t)NSLOTS / 2) { need_sync = true; must_finish.push_back(p); }
      else kept++;
    }
  }
  if (need_sync) {  // everything enqueued so far has run after this: a report of must_finish that is still missing never comes
    MaybeReleaseGil nogil;
    c10::hip::HIPGuard g(st.index);
    (void)hipDeviceSynchronize();
  }
  {
    std::lock_guard<std::mutex> lk(st.mu);
    std::deque<std::shared_ptr<PendingRec>> keep;
    for (std::shared_ptr<PendingRec>& p : st.pending) {
      const int rc = poll(*p);
      if (rc == MGS_PENDING) {
        if (std::find(must_finish.begin(), must_finish.end(), p) != must_finish.end()) {
          if (failed.empty()) failed = "a rasterizer forward finished without reporting its instance count";
        } else {
          keep.push_back(p);  // (enqueued by another thread in the meantime, or not yet due)
        }
        continue;
      }
      std::string f = account(st, *p, rc, &warnings);
      if (failed.empty()) failed = f;
    }
    st.pending.swap(keep);
  }
  if (deferred) deferred->insert(deferred->end(), warnings.begin(), warnings.end());
  else for (const std::string& w : warnings) warn_runtime(w);
  if (!failed.empty()) fail(failed);
}

bool is_capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }  // when in doubt: the Python path
  return cs != hipStreamCaptureStatusNone;
}

inline size_t up256(size_t n) { return (n + 255) & ~(size_t)255; }

bool supported_F(int64_t F) { return F == 3 || F == 4 || F == 8 || F == 16 || F == 32 || F == 64; }

// Float offsets of the backward's single allocation (manigaussian_amd/_C.py _grad_layout):
// [scratch | dL_dcolors | dL_dfeature | means3D | opacity | sh | scales | rotations | cov3D | means2D | pad]
struct GradLayout { int64_t scr, col, feat, m3, op, sh, sc, rot, cov, m2, total; size_t accum_bytes; };
GradLayout grad_layout(int64_t P, int64_t M, int64_t F) {
  GradLayout g;
  const int64_t scratch_f = (int64_t)((mgs_backward_scratch_bytes((int)P, (int)M, (int)F) + 3) / 4);
  int64_t o = 0;
  g.scr = o; o += scratch_f;
  g.col = o; o += 3 * P;
  g.feat = o; o += F * P;
  g.m3 = o; o += 3 * P;
  g.op = o; o += P;
  g.sh = o; o += 3 * M * P;
  g.sc = o; o += 3 * P;
  g.rot = o; o += 4 * P;
  g.cov = o; o += 6 * P;
  g.m2 = o; o += 3 * P;
  o += 4;
  g.total = o;
  g.accum_bytes = (size_t)(((scratch_f + 3 * P + F * P) * 4 + 15) / 16 * 16);
  return g;
}

const char* kBackwardTwice =
    "Trying to backward through the graph a second time (or directly access saved tensors after they have already been freed). "
    "Saved intermediate values of the graph are freed when you call .backward() or autograd.grad(). Specify retain_graph=True "
    "if you need to backward through the graph a second time or if you need to access saved tensors after calling backward.";

// ---- the autograd node: what _RasterizeGaussians.backward is in the reference (RAST/diff_gaussian_rasterization/__init__.py:
// 104-164), called by the engine without entering Python -------------------------------------------------------------------
struct MgsRasterizeBackward : public Node {
  MgsRasterArgs a;                       // the forward's arguments, reused as they were
  std::shared_ptr<PendingRec> pending;   // its outstanding device report (nullptr: settled in the forward)
  std::vector<SavedVariable> saved;      // the reference's saved inputs: colors, feature, means3D, scales, rotations, cov3D, sh
  at::Tensor opacities, bg, viewmatrix, projmatrix, campos;  // kept alive (the argument struct holds their addresses)
  at::Tensor radii, ws, binning2, grad_buffer;
  std::shared_ptr<Lease> lease;          // the budget charge of a worst-case workspace (released with `ws`)
  // the images, weakly (they own this node through their grad_fn): a recovery re-renders into them if they still live
  c10::weak_intrusive_ptr<c10::TensorImpl> out_color{c10::intrusive_ptr<c10::TensorImpl>()}, out_feat{c10::intrusive_ptr<c10::TensorImpl>()};
  int device_index = 0;
  int32_t num_rendered = -1;             // the reference's count if the forward waited for it, else -1
  int64_t P = 0, M = 0, F = 0, H = 0, W = 0;
  bool include_feature = false;
  bool released = false;
  std::mutex mu;

  std::string name() const override { return "MgsRasterizeBackward"; }
  void release_variables() override {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& s : saved) s.reset_data();
    saved.clear();
    opacities = bg = viewmatrix = projmatrix = campos = radii = ws = binning2 = grad_buffer = at::Tensor();
    lease.reset();
    released = true;
  }
  variable_list apply(variable_list&& grads) override;
  variable_list apply_locked(variable_list&& grads, std::vector<std::string>* warnings);
  void settle(DeviceState& st, hipStream_t stream, std::vector<std::string>* warnings);
  void recover(DeviceState& st, hipStream_t stream, bool handshake_only);
};

// The asynchronous forward of this backward outgrew its workspace and the report is in: bin and render it AGAIN with room
// (capacity from the count the device reported, worst-case chunk pool: cannot overflow) into the same output tensors, so that the
// backward runs on a complete state (manigaussian_amd/_C.py recover_forward).
void MgsRasterizeBackward::recover(DeviceState& st, hipStream_t stream, bool handshake_only) {
  if (handshake_only) {
    // the scene did not outgrow anything: a workgroup of the preprocess launch gave up waiting for workgroup 0's zeroed tables.
    // Same workspace, tables zeroed by a launch of their own this time (no workgroup waits for another).
    a.opt.set = 1;
    a.opt.table_init = 1;
  } else {
    const int64_t R = std::max<int64_t>(pending->num_rendered, 0);
    const int64_t cap = R + R / 4 + 4096;
    const int Fl = include_feature ? (int)F : 0;
    const size_t nbytes = mgs_binning_bytes2((int)cap, 0, (int)W, (int)H, Fl);
    binning2 = at::empty({(int64_t)nbytes}, ws.options());
    a.binning = binning2.data_ptr();
    a.binning_bytes = nbytes;
    a.binning_capacity = (int32_t)cap;
    a.chunk_pool = 0;
  }
  a.async_forward = 0;
  a.bwd_accum = nullptr;  // the first run's preprocess zeroed the accumulators; nothing touched them since
  a.bwd_accum_bytes = 0;
  uint64_t* slot;
  uint32_t tag;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    slot = st.take_slot(&tag);
  }
  a.status_tag = tag;
  int32_t nr = 0;
  const auto f32 = radii.options().dtype(at::kFloat);
  auto alive = [&](const c10::weak_intrusive_ptr<c10::TensorImpl>& w, at::IntArrayRef shape) {
    c10::intrusive_ptr<c10::TensorImpl> p = w.lock();
    return p ? at::Tensor(std::move(p)) : at::empty(shape, f32);  // (nobody will read a fresh one; the kernels need a target)
  };
  at::Tensor oc = alive(out_color, {3, H, W});
  at::Tensor of = include_feature ? alive(out_feat, {F, H, W}) : at::Tensor();
  {
    MaybeReleaseGil nogil;
    check_rc(mgs_rasterize_forward(&a, radii.data_ptr<int32_t>(), oc.data_ptr<float>(),
                                   include_feature ? of.data_ptr<float>() : nullptr, &nr, slot, stream),
             "rasterizer forward (re-run after a workspace overflow)");
  }
  auto np = std::make_shared<PendingRec>();
  np->a = a; np->slot = slot; np->key = pending->key;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    pending->recovered = true;
    st.pending.push_back(np);
  }
  pending = np;
  num_rendered = nr;
  counters().recovered++;
}

// Backward entry: what is known about this backward's forward?  (manigaussian_amd/_C.py _settle)
void MgsRasterizeBackward::settle(DeviceState& st, hipStream_t stream, std::vector<std::string>* warnings) {
  if (!pending) return;
  int rc;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    rc = poll(*pending);
  }
  if (rc == MGS_NEED_CAPACITY || rc == MGS_RETRY_TABLE_INIT) {
    if (cfg().policy.load() == POLICY_RAISE) {
      drain(st, false, warnings);  // folds the report into the marks and raises ... unless an earlier drain already did
      fail(rc == MGS_NEED_CAPACITY
               ? "the asynchronous rasterizer forward of this backward outgrew its workspace (the scene grew past the head-room over "
                 "earlier calls of its shape): its images are incomplete and the step is lost; the next call of the shape gets a "
                 "larger workspace (overflow policy 'raise')"
               : std::string(kHandshake) + ": its images are incomplete and the step is lost (overflow policy 'raise')");
    }
    drain(st, false, warnings);  // learn + warn (this pending is recoverable and its backward has not been enqueued: no raise)
    recover(st, stream, rc == MGS_RETRY_TABLE_INIT);
    return;
  }
  if (rc != MGS_OK && rc != MGS_PENDING) drain(st, false, warnings);
  if (rc == MGS_PENDING) {
    std::lock_guard<std::mutex> lk(st.mu);
    pending->backward_enqueued = true;  // too late to repair: if this forward overflowed, the next drain raises
  }
}

// The node's mutex is never held while a warning is issued (warn_runtime takes the GIL; a Python thread that holds the GIL may
// be waiting for this mutex in release_variables): apply_locked collects them, apply issues them after the lock is gone.
variable_list MgsRasterizeBackward::apply(variable_list&& grads) {
  std::vector<std::string> warnings;
  variable_list out;
  std::exception_ptr err;
  try {
    out = apply_locked(std::move(grads), &warnings);
  } catch (...) {
    err = std::current_exception();
  }
  for (const std::string& w : warnings) warn_runtime(w);
  if (err) std::rethrow_exception(err);
  return out;
}

variable_list MgsRasterizeBackward::apply_locked(variable_list&& grads, std::vector<std::string>* warnings) {
  SegClock clk;
  std::lock_guard<std::mutex> lk(mu);
  TORCH_CHECK(!released, kBackwardTwice);
  variable_list out(9);
  at::Tensor g_color = grads.size() > 0 ? grads[0] : at::Tensor();
  at::Tensor g_feat = grads.size() > 1 ? grads[1] : at::Tensor();
  if (!g_color.defined() && !g_feat.defined()) return out;  // nothing flows back
  for (auto& s : saved) (void)s.unpack(shared_from_this());  // raises if a saved input was modified in place since the forward
  c10::hip::HIPGuard guard(device_index);
  const auto f32 = radii.options().dtype(at::kFloat);
  if (!g_color.defined()) g_color = at::zeros({3, H, W}, f32);
  if (g_color.scalar_type() != at::kFloat) fail("expected scalar type Float for dL_dout_color");
  if (!g_color.is_contiguous()) g_color = g_color.contiguous();
  if (include_feature) {
    if (!g_feat.defined()) g_feat = at::zeros({F, H, W}, f32);
    if (g_feat.scalar_type() != at::kFloat) fail("expected scalar type Float for dL_dout_language_feature");
    if (!g_feat.is_contiguous()) g_feat = g_feat.contiguous();
  }
  hipStream_t stream = c10::hip::getCurrentHIPStream(device_index).stream();
  DeviceState& st = state(device_index);
  clk.lap(SB_UNPACK);
  settle(st, stream, warnings);
  clk.lap(SB_SETTLE);
  const GradLayout L = grad_layout(P, M, include_feature ? F : 0);
  const bool prezeroed = grad_buffer.defined() && grad_buffer.numel() == L.total;
  at::Tensor flat = prezeroed ? grad_buffer : at::empty({L.total}, f32);
  grad_buffer = at::Tensor();  // pre-zeroed for ONE backward; a second one (retain_graph) allocates and fills
  float* base = flat.data_ptr<float>();
  a.accum_prezeroed = prezeroed ? 1 : 0;
  clk.lap(SB_ALLOC);
  check_rc(mgs_rasterize_backward(&a, num_rendered, radii.data_ptr<int32_t>(), g_color.data_ptr<float>(),
                                  include_feature ? g_feat.data_ptr<float>() : nullptr, base + L.m2, nullptr, base + L.op,
                                  base + L.col, include_feature ? base + L.feat : nullptr, base + L.m3, base + L.cov,
                                  M ? base + L.sh : nullptr, base + L.sc, base + L.rot, base + L.scr,
                                  (size_t)(L.col - L.scr) * 4, stream),
           "rasterize_gaussians_backward");
  clk.lap(SB_LIBRARY);
  // order of the forward's inputs (reference: __init__.py:151-162): means3D, means2D, sh, colors_precomp, language_feature,
  // opacities, scales, rotations, cov3D_precomp -- only what somebody asked for
  auto view = [&](int i, at::IntArrayRef sizes, at::IntArrayRef strides, int64_t off) {
    if (task_should_compute_output(i)) out[i] = flat.as_strided(sizes, strides, off);
  };
  view(0, {P, 3}, {3, 1}, L.m3);
  view(1, {P, 3}, {3, 1}, L.m2);
  if (M) view(2, {P, M, 3}, {3 * M, 3, 1}, L.sh);
  view(3, {P, 3}, {3, 1}, L.col);
  if (include_feature) view(4, {P, F}, {F, 1}, L.feat);
  view(5, {P, 1}, {1, 1}, L.op);
  view(6, {P, 3}, {3, 1}, L.sc);
  view(7, {P, 4}, {4, 1}, L.rot);
  view(8, {P, 6}, {6, 1}, L.cov);
  clk.lap(SB_VIEWS);
  counters().backwards++;
  return out;
}

// a tensor the C ABI can take as it is: float32, on `dev`, contiguous -- or empty (a NULL pointer, like the reference's
// .data<float>() of torch.Tensor([]))
inline bool plain(const at::Tensor& t, const c10::Device& dev) {
  if (!t.defined()) return false;
  if (t.numel() == 0) return true;
  return t.scalar_type() == at::kFloat && t.device() == dev && t.is_contiguous();
}
inline const float* fptr(const at::Tensor& t) { return t.numel() == 0 ? nullptr : t.data_ptr<float>(); }

struct WorstCase { int64_t P, W, H, F, budget; bool ok; int64_t cap, pool; };

// GaussianRasterizer.forward -> _RasterizeGaussians.apply (RAST/diff_gaussian_rasterization/__init__.py:21-103), the hot path.
// Returns None when the call is not one this binding handles; the caller then takes manigaussian_amd/_C.py.
py::object rasterize(const at::Tensor& means3D, const at::Tensor& means2D, const at::Tensor& sh, const at::Tensor& colors,
                     const at::Tensor& feat, const at::Tensor& opacities, const at::Tensor& scales, const at::Tensor& rotations,
                     const at::Tensor& cov3D, const at::Tensor& bg, const at::Tensor& viewmatrix, const at::Tensor& projmatrix,
                     const at::Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                     int64_t degree, bool prefiltered, bool debug, bool include_feature) {
  SegClock clk;
  Config& C = cfg();
  const int mode = C.mode.load();
  auto decline = [&]() { counters().declined++; return py::none(); };
  if (!C.enabled.load() || debug || prefiltered || mode == MODE_BLOCKING) return decline();
  if (!means3D.defined() || !means3D.is_cuda() || means3D.dim() != 2 || means3D.size(1) != 3) return decline();  // (Python raises)
  const int64_t P = means3D.size(0);
  if (P == 0 || H <= 0 || W <= 0) return decline();
  const c10::Device dev = means3D.device();
  if (!plain(means3D, dev) || !plain(sh, dev) || !plain(colors, dev) || !plain(opacities, dev) || !plain(scales, dev) ||
      !plain(rotations, dev) || !plain(cov3D, dev) || !plain(bg, dev) || !plain(viewmatrix, dev) || !plain(projmatrix, dev) ||
      !plain(campos, dev) || opacities.numel() == 0 || bg.numel() == 0 || !means2D.defined())
    return decline();
  const int64_t M = sh.numel() != 0 ? (sh.dim() == 3 ? sh.size(1) : -1) : 0;
  if (M < 0) return decline();
  int64_t F = 0;
  if (include_feature) {
    if (!plain(feat, dev) || feat.dim() != 2 || feat.size(0) != P || !supported_F(feat.size(1)) ||
        (reinterpret_cast<uintptr_t>(feat.data_ptr()) & 15u))
      return decline();  // padded widths, odd offsets: the Python path copies
    F = feat.size(1);
  }
  const int di = dev.index();
  clk.lap(SG_CHECKS);
  c10::hip::HIPGuard guard(di);
  hipStream_t stream = c10::hip::getCurrentHIPStream(di).stream();
  if (is_capturing(stream)) return decline();
  DeviceState& st = state(di);
  clk.lap(SG_STREAM);
  // reports of forwards the Python shim enqueued; and, once per device, the shim's own state (its pinned status ring cannot
  // be allocated later, inside a HIP-graph capture -- which is the shim's job)
  if (g_py_drain && ((di < MAX_DEVICES && g_py_pending[di].load() > 0) || !st.python_side_built)) {
    (*g_py_drain)(di);
    st.python_side_built = true;
  }
  drain(st, false);  // reports of earlier forwards that have arrived: learn their counts, raise if one overflowed
  clk.lap(SG_DRAIN);

  MgsOptions opt;
  double head_inst;
  int64_t safe_bytes;
  {
    std::lock_guard<std::mutex> lk(C.mu);
    opt = C.opt; head_inst = C.head_inst; safe_bytes = C.safe_bytes;
  }
  (void)head_inst;
  if (safe_bytes < 0) {  // the default budget: 1/32 of this device's memory, at least 1 GB
    if (st.auto_safe_bytes == 0) {
      size_t total = 0;
      if (hipDeviceTotalMem(&total, di) != hipSuccess) { (void)hipGetLastError(); total = 0; }
      st.auto_safe_bytes = std::max<int64_t>((int64_t)1 << 30, (int64_t)(total / 32));
    }
    safe_bytes = st.auto_safe_bytes;
  }
  const Key key{0, (int32_t)P, (int32_t)W, (int32_t)H, (int32_t)F, opt.tight_bins};
  const int64_t T = ((W + 15) / 16) * ((H + 15) / 16);
  // worst case: every Gaussian in every tile, every chunk of every block visited
  static thread_local WorstCase wc{-1, 0, 0, 0, 0, false, 0, 0};
  if (!(wc.P == P && wc.W == W && wc.H == H && wc.F == F && wc.budget == safe_bytes)) {
    const int64_t cap_worst = P * T;
    const bool ok = cap_worst < ((int64_t)1 << 30) &&
                    (int64_t)mgs_binning_bytes2((int)cap_worst, 0, (int)W, (int)H, (int)F) <= safe_bytes;
    wc = WorstCase{P, W, H, F, safe_bytes, ok, cap_worst, ok ? mgs_chunk_pool_max((int)cap_worst, (int)W, (int)H) : 0};
  }
  // The worst-case workspace is taken only while the worst cases live forwards of this device still hold, plus this one, stay
  // within the budget (round 5 tested the budget per call: V forwards before the first backward pinned V x 4 GB at
  // configs[2]) -- and only if the allocator can give it; otherwise the shape goes by its marks, like any shape whose
  // worst case does not fit (the reference never allocates more than the count needs, rasterize_points.cu:27-33,84-89).
  const auto u8 = at::TensorOptions().dtype(at::kByte).device(dev);
  const auto f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
  const size_t gb = up256(mgs_geom_bytes((int)P, (int)M, (int)W, (int)H)), ib = up256(mgs_img_bytes((int)W, (int)H));
  const bool want_grad = torch::autograd::compute_requires_grad(means3D, means2D, sh, colors, feat, opacities, scales, rotations, cov3D);
  at::AutoDispatchBelowADInplaceOrView below_autograd;  // plain tensors from here on; the node is attached by hand
  at::Tensor ws;
  std::shared_ptr<Lease> lease;
  size_t bb = 0;
  bool worst = mode != MODE_ASYNC && wc.ok;
  if (worst) {
    bb = mgs_binning_bytes2((int)wc.cap, (int)wc.pool, (int)W, (int)H, (int)F);
    if (st.held_bytes.load() + (int64_t)bb > safe_bytes) {
      worst = false;
    } else {
      try {
        ws = at::empty({(int64_t)(gb + ib + bb)}, u8);
        if (want_grad) lease = std::make_shared<Lease>(&st, (int64_t)bb);  // (without a backward the workspace dies with this call)
      } catch (const c10::OutOfMemoryError&) {
        worst = false;
        ws = at::Tensor();
      }
    }
    if (!worst) counters().budget_fallbacks++;
  }
  int64_t cap = 0, pool = 0, mark_R = 0;
  bool have_guess = worst;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    if (worst) { cap = wc.cap; pool = wc.pool; }
    else have_guess = st.guess(key, &cap, &pool);
    auto it = st.marks.find(key);
    if (it != st.marks.end()) mark_R = it->second.R;
  }
  const bool lazy = have_guess && (mode == MODE_ASYNC || (mode == MODE_SAFE && worst));
  if (mode == MODE_ASYNC && !lazy) return decline();  // first calls of a shape in async mode: the Python path learns the marks
  if (!lazy) {  // wait for the preprocess: capacity from the marks, worst-case pool for that capacity (cannot overflow)
    cap = mark_R ? mark_R + mark_R / 4 + 4096 : 4 * P + 4096;
    pool = 0;
  }
  if (cap >= ((int64_t)1 << 31)) return decline();

  clk.lap(SG_SIZES);
  // ONE allocation for the three opaque workspaces [geom | img | binning] (each a multiple of 256 bytes)
  if (!worst) {
    // (+ room for the forward preprocess to write the tile keys itself: no bin scatter launch, include/mgsplat.h
    //  mgs_binning_direct_extra; a worst-case capacity, above, covers them by itself)
    bb = up256(mgs_binning_bytes2((int)cap, (int)pool, (int)W, (int)H, (int)F)) + mgs_binning_direct_extra((int)P, 1, (int)W, (int)H);
    ws = at::empty({(int64_t)(gb + ib + bb)}, u8);
  }
  at::Tensor out_color, out_feat;
  if (include_feature) {
    at::Tensor out = at::empty({3 + F, H, W}, f32);
    out_color = out.narrow(0, 0, 3);
    out_feat = out.narrow(0, 3, F);
  } else {
    out_color = at::empty({3, H, W}, f32);
    out_feat = at::zeros({1}, f32);  // rasterize_points.cu:71-79: a [1] placeholder
  }
  at::Tensor radii = at::empty({P}, at::TensorOptions().dtype(at::kInt).device(dev));
  at::Tensor grad_buffer;
  GradLayout GL{};
  if (want_grad) {
    GL = grad_layout(P, M, F);
    grad_buffer = at::empty({GL.total}, f32);
  }
  clk.lap(SG_ALLOC);

  MgsRasterArgs a;
  std::memset(&a, 0, sizeof(a));
  a.P = (int32_t)P; a.D = (int32_t)degree; a.M = (int32_t)M; a.F = (int32_t)F; a.W = (int32_t)W; a.H = (int32_t)H;
  a.tanfovx = (float)tanfovx; a.tanfovy = (float)tanfovy; a.scale_modifier = (float)scale_modifier;
  a.prefiltered = 0; a.debug = 0; a.include_feature = include_feature ? 1 : 0;
  a.background = fptr(bg); a.means3D = fptr(means3D); a.shs = fptr(sh); a.colors_precomp = fptr(colors);
  a.language_feature = include_feature ? fptr(feat) : nullptr;
  a.opacities = fptr(opacities); a.scales = fptr(scales); a.rotations = fptr(rotations); a.cov3D_precomp = fptr(cov3D);
  a.viewmatrix = fptr(viewmatrix); a.projmatrix = fptr(projmatrix); a.campos = fptr(campos);
  char* wsp = reinterpret_cast<char*>(ws.data_ptr());
  a.geom = wsp; a.geom_bytes = gb; a.img = wsp + gb; a.img_bytes = ib; a.binning = wsp + gb + ib; a.binning_bytes = bb;
  a.binning_capacity = (int32_t)cap; a.chunk_pool = (int32_t)pool;
  a.async_forward = lazy ? 1 : 0;
  a.opt = opt;
  if (opt.seg == 2048 && mark_R > 8192 * T) a.opt.seg = 4096;  // long per-tile lists (manigaussian_amd/_lib.py auto_seg)
  if (opt.bin_mode == 2 && mark_R > 24576 * T) {               // very long ones: segment sort + rank merge (same place)
    a.opt.bin_mode = 1;
    if (opt.seg == 2048) a.opt.seg = 4096;
  }
  if (want_grad) {
    a.bwd_accum = grad_buffer.data_ptr();
    a.bwd_accum_bytes = GL.accum_bytes;
  }
  uint64_t* slot;
  uint32_t tag;
  {
    std::lock_guard<std::mutex> lk(st.mu);
    slot = st.take_slot(&tag);
  }
  a.status_tag = tag;
  int32_t nr = 0;
  int rc;
  float* feat_ptr = include_feature ? out_feat.data_ptr<float>() : nullptr;
  clk.lap(SG_ARGS);
  if (lazy) {
    rc = mgs_rasterize_forward(&a, radii.data_ptr<int32_t>(), out_color.data_ptr<float>(), feat_ptr, &nr, slot, stream);
  } else {
    py::gil_scoped_release nogil;  // the call polls a pinned word until the preprocess has reported
    rc = mgs_rasterize_forward(&a, radii.data_ptr<int32_t>(), out_color.data_ptr<float>(), feat_ptr, &nr, slot, stream);
    counters().waited++;
  }
  std::shared_ptr<PendingRec> pending;
  at::Tensor binning2;
  if (rc == MGS_NEED_CAPACITY) {  // (waiting path) first call of the shape, or the scene grew: bin + render again with room
    const int64_t binned = (int64_t)(uint32_t)(*(volatile uint64_t*)slot);
    const int64_t cap2 = (int64_t)nr + nr / 4 + 4096;  // (nr: the reference's 3-sigma-rect count, at least the instances binned)
    const size_t nb = mgs_binning_bytes2((int)cap2, 0, (int)W, (int)H, (int)F);
    binning2 = at::empty({(int64_t)nb}, u8);
    a.binning = binning2.data_ptr(); a.binning_bytes = nb; a.binning_capacity = (int32_t)cap2; a.chunk_pool = 0;
    check_rc(mgs_rasterize_forward_render(&a, nr, radii.data_ptr<int32_t>(), out_color.data_ptr<float>(), feat_ptr, stream),
             "rasterize_gaussians");
    std::lock_guard<std::mutex> lk(st.mu);
    st.learn(key, binned, -1, false);
    counters().retried++;
  } else {
    check_rc(rc, "rasterize_gaussians");
    pending = std::make_shared<PendingRec>();
    pending->a = a; pending->slot = slot; pending->key = key;
    pending->recoverable = want_grad;
    std::lock_guard<std::mutex> lk(st.mu);
    st.pending.push_back(pending);
  }
  counters().forwards++;
  clk.lap(SG_LIBRARY);
  if (want_grad) {
    std::shared_ptr<MgsRasterizeBackward> node(new MgsRasterizeBackward(), torch::autograd::deleteNode);
    node->set_next_edges(torch::autograd::collect_next_edges(means3D, means2D, sh, colors, feat, opacities, scales, rotations, cov3D));
    node->a = a; node->pending = pending; node->device_index = di;
    node->num_rendered = lazy ? -1 : nr;
    node->P = P; node->M = M; node->F = F; node->H = H; node->W = W; node->include_feature = include_feature;
    node->saved.reserve(7);
    for (const at::Tensor* t : {&colors, &feat, &means3D, &scales, &rotations, &cov3D, &sh}) node->saved.emplace_back(*t, false);
    node->opacities = opacities; node->bg = bg; node->viewmatrix = viewmatrix; node->projmatrix = projmatrix; node->campos = campos;
    node->radii = radii; node->ws = ws; node->binning2 = binning2; node->grad_buffer = grad_buffer;
    node->lease = lease;
    node->out_color = c10::weak_intrusive_ptr<c10::TensorImpl>(out_color.getIntrusivePtr());
    node->out_feat = c10::weak_intrusive_ptr<c10::TensorImpl>(out_feat.getIntrusivePtr());
    torch::autograd::set_history(out_color, node);
    torch::autograd::set_history(out_feat, node);
  }
  clk.lap(SG_NODE);
  py::object ret = py::make_tuple(out_color, out_feat, radii);
  clk.lap(SG_RETURN);
  return ret;
}

// ---- configuration and the marks, driven by manigaussian_amd/_state.py and _lib.py ----------------------------------------
Key key_of(const py::tuple& t) {
  if (t.size() == 5) return Key{0, t[0].cast<int>(), t[1].cast<int>(), t[2].cast<int>(), t[3].cast<int>(), t[4].cast<int>()};
  if (t.size() == 7) return Key{t[1].cast<int>(), t[2].cast<int>(), t[3].cast<int>(), t[4].cast<int>(), t[5].cast<int>(), t[6].cast<int>()};
  throw py::key_error("a marks key is (P, W, H, F, tight_bins) or ('views', V, P, W, H, F, tight_bins)");
}
py::tuple tuple_of(const Key& k) {
  if (k.V == 0) return py::make_tuple(k.P, k.W, k.H, k.F, k.tight);
  return py::make_tuple("views", k.V, k.P, k.W, k.H, k.F, k.tight);
}

py::object marks_get(int dev, const py::tuple& key) {
  DeviceState& st = state(dev);
  std::lock_guard<std::mutex> lk(st.mu);
  auto it = st.marks.find(key_of(key));
  if (it == st.marks.end()) return py::none();
  py::list l;
  l.append(it->second.R);
  l.append(it->second.chunks < 0 ? py::object(py::none()) : py::object(py::int_(it->second.chunks)));
  return std::move(l);
}
void marks_set(int dev, const py::tuple& key, int64_t R, py::object chunks) {
  DeviceState& st = state(dev);
  std::lock_guard<std::mutex> lk(st.mu);
  Mark& m = st.marks[key_of(key)];
  m.R = R;
  m.chunks = chunks.is_none() ? -1 : chunks.cast<int64_t>();
}
bool marks_del(int dev, const py::tuple& key) {
  DeviceState& st = state(dev);
  std::lock_guard<std::mutex> lk(st.mu);
  return st.marks.erase(key_of(key)) != 0;
}
py::list marks_keys(int dev) {
  DeviceState& st = state(dev);
  std::lock_guard<std::mutex> lk(st.mu);
  py::list l;
  for (auto& kv : st.marks) l.append(tuple_of(kv.first));
  return l;
}

void configure(int mode, int policy, double head_inst, double head_chunks, int64_t safe_bytes) {
  Config& C = cfg();
  std::lock_guard<std::mutex> lk(C.mu);
  C.mode = mode; C.policy = policy; C.head_inst = head_inst; C.head_chunks = head_chunks; C.safe_bytes = safe_bytes;
}
void set_options(int tight_bins, int fast_exp, int exact_cull, int bin_mode, int seg, int gm_waves, int dbg, int table_init) {
  Config& C = cfg();
  std::lock_guard<std::mutex> lk(C.mu);
  C.opt.set = 1; C.opt.tight_bins = tight_bins; C.opt.fast_exp = fast_exp; C.opt.exact_cull = exact_cull; C.opt.bin_mode = bin_mode;
  C.opt.seg = seg; C.opt.gm_waves = gm_waves; C.opt.dbg = dbg; C.opt.table_init = table_init;
}
bool set_enabled(bool on) { return cfg().enabled.exchange(on); }

void check_status(int dev, bool wait) {  // dev < 0: every device used so far
  std::vector<DeviceState*> all;
  {
    std::lock_guard<std::mutex> lk(g_states_mu);
    for (auto& s : g_states)
      if (s && (dev < 0 || s->index == dev)) all.push_back(s.get());
  }
  for (DeviceState* s : all) drain(*s, wait);
}
int pending_count(int dev) {
  DeviceState& st = state(dev);
  std::lock_guard<std::mutex> lk(st.mu);
  return (int)st.pending.size();
}
void set_python_drain(py::object fn) {
  delete g_py_drain;
  g_py_drain = fn.is_none() ? nullptr : new py::object(std::move(fn));
}
void set_python_pending(int dev, int n) {
  if (dev >= 0 && dev < MAX_DEVICES) g_py_pending[dev] = n;
}
// the ctypes shim's worst-case workspaces are charged against the same per-device sum (manigaussian_amd/_state.py hold())
int64_t hold_add(int dev, int64_t delta) { return state(dev).held_bytes += delta; }
int64_t held_bytes(int dev) { return state(dev).held_bytes.load(); }

py::dict profile_read(bool reset) {
  py::dict d;
  for (int i = 0; i < SG_COUNT; i++) {
    d[kSegNames[i]] = py::make_tuple(g_seg_ns[i].load(), g_seg_n[i].load());
    if (reset) { g_seg_ns[i] = 0; g_seg_n[i] = 0; }
  }
  return d;
}

py::dict counters_dict(bool reset) {
  Counters& c = counters();
  py::dict d;
  d["forwards"] = c.forwards.load(); d["backwards"] = c.backwards.load(); d["declined"] = c.declined.load();
  d["waited"] = c.waited.load(); d["retried"] = c.retried.load(); d["recovered"] = c.recovered.load();
  d["budget_fallbacks"] = c.budget_fallbacks.load();
  if (reset) { c.forwards = 0; c.backwards = 0; c.declined = 0; c.waited = 0; c.retried = 0; c.recovered = 0; c.budget_fallbacks = 0; }
  return d;
}

}  // namespace

PYBIND11_MODULE(_mgs_torch, m) {
  m.doc() = "compiled autograd binding of libmgsplat.so (include/mgsplat.h); see manigaussian_amd/csrc/mgs_torch.cpp";
  if (mgs_abi_version() != MGS_ABI_VERSION)
    throw std::runtime_error("libmgsplat ABI version " + std::to_string(mgs_abi_version()) + " != the binding's " +
                             std::to_string(MGS_ABI_VERSION) + "; rebuild");
  mgs_options_default(&cfg().opt);
  m.attr("ABI_VERSION") = MGS_ABI_VERSION;
  m.def("build_id", []() { return std::string(mgs_build_id()); });
  m.def("rasterize", &rasterize);
  m.def("configure", &configure);
  m.def("set_options", &set_options);
  m.def("set_enabled", &set_enabled);
  m.def("enabled", []() { return cfg().enabled.load(); });
  m.def("check_status", &check_status, py::arg("device") = -1, py::arg("wait") = true);
  m.def("pending_count", &pending_count);
  m.def("marks_get", &marks_get);
  m.def("marks_set", &marks_set);
  m.def("marks_del", &marks_del);
  m.def("marks_keys", &marks_keys);
  m.def("set_python_drain", &set_python_drain);
  m.def("set_python_pending", &set_python_pending);
  m.def("hold_add", &hold_add);
  m.def("held_bytes", &held_bytes);
  m.def("counters", &counters_dict, py::arg("reset") = false);
  m.def("set_profile", [](bool on) { return g_profile.exchange(on); });
  m.def("profile_read", &profile_read, py::arg("reset") = true);
}
