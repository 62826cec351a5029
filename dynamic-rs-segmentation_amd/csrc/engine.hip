// Step-level C ABI: the three `sess.run` call shapes of the reference as three entry points of libdrs_hip.so.
//
//   train  sess.run([optimizer, loss, pred_up], is_training=True)    isprs_dilated_random.py:1750-1752  -> drs_train_step
//   infer  sess.run([pred_up, logits],         is_training=False)   isprs_dilated_random.py:1274-1275  -> drs_forward
//   val    sess.run(pred_up,                   is_training=False)   isprs_dilated_random.py:1588       -> drs_forward
//
// A `drs_net` owns what the TensorFlow graph owned: the net table selected by `net_type` (the if-chain isprs:1660-1680; builders
// isprs:761-1086, coffee:665-841, contest:574-641), the variable layout (TF scope names), and the order in which the op-level
// entry points of this library (conv_mfma.hip, pointwise.hip) are enqueued for a forward pass and for a training step --
// exactly the sequence of dynamic-rs-segmentation_amd/net.py (exact-fp32 arithmetic), which stays as the op-level host mirror
// and is held bitwise equal to this file by tests/test_gpu_engine.py.  It owns NO device memory: the caller allocates every
// buffer drs_net_buffer_info lists (any allocator) and binds it; nothing here allocates, frees or synchronises the device,
// except drs_params_get / drs_params_set, which copy to / from host memory and wait for that copy.
// Data parallelism stays with the driver: every sum that has to run over all ranks goes through the all-reduce callback
// (drs_net_set_comm), at the same points and on the same buffers as net.py does it through torch.distributed.
#include "drs_common.hpp"
#include "../../include/drs.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

__attribute__((visibility("hidden"))) int drs_step_prep(const StepPrepArgs& a, hipStream_t stream);      // conv_mfma.hip
__attribute__((visibility("hidden"))) int drs_rccl_all_reduce_sum(void* comm, void* ptr, size_t count, int dtype, hipStream_t stream);      // rccl_comm.hip

namespace {

// ------------------------------------------------------------------------------------------------ net tables
// (scope, k, c_in (-1 = image bands), c_out, rate, k_dim (squeeze layers only))
struct ConvRow { const char* name; int k, cin, cout, rate, kdim; };
enum Topo { CHAIN = 0, DENSE = 1, SQUEEZE = 2 };
struct NetTable {
  const char* net_type;
  float alpha;                 // max(alpha x, x): 0 = ReLU, 0.1 = leaky ReLU (isprs:620-621)
  int max_pool;                // 3x3 / stride 1 max-pool after every block (the *_grsl* nets, isprs:745-746)
  int avg_pool[8];             // or: k of a k x k stride-1 SAME average pool after block i (0 = none), isprs:818-854
  Topo topo;
  int c_last;
  int nconv;
  ConvRow convs[8];
  int se_after[3];             // blocks followed by a squeeze-and-excitation layer (isprs:1042-1050), -1 = none
};

const NetTable kTables[] = {
    {"dilated_icpr_original", 0.f, 0, {0}, CHAIN, 256, 6,
     {{"main_conv1", 5, -1, 64, 1, 0}, {"main_conv2", 5, 64, 64, 1, 0}, {"main_conv3", 4, 64, 128, 2, 0}, {"main_conv4", 4, 128, 128, 2, 0},
      {"main_conv5", 3, 128, 256, 4, 0}, {"main_conv6", 3, 256, 256, 4, 0}}, {-1, -1, -1}},
    {"dilated_grsl", 0.1f, 1, {0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 128, 3, 0}, {"conv4", 4, 128, 128, 4, 0}, {"conv5", 3, 128, 256, 5, 0},
      {"conv6", 3, 256, 256, 6, 0}}, {-1, -1, -1}},
    {"dilated_grsl_rate8", 0.1f, 1, {0}, CHAIN, 256, 8,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 128, 3, 0}, {"conv4", 4, 128, 128, 4, 0}, {"conv5", 3, 128, 192, 5, 0},
      {"conv6", 3, 192, 192, 6, 0}, {"conv7", 3, 192, 256, 7, 0}, {"conv8", 3, 256, 256, 8, 0}}, {-1, -1, -1}},
    {"dilated_icpr_rate6", 0.f, 0, {0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 128, 3, 0}, {"conv4", 4, 128, 128, 4, 0}, {"conv5", 3, 128, 256, 5, 0},
      {"conv6", 3, 256, 256, 6, 0}}, {-1, -1, -1}},
    {"dilated_icpr_rate6_small", 0.f, 0, {0}, CHAIN, 128, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 64, 3, 0}, {"conv4", 4, 64, 128, 4, 0}, {"conv5", 3, 128, 128, 5, 0},
      {"conv6", 3, 128, 128, 6, 0}}, {-1, -1, -1}},
    {"dilated_icpr_rate6_nodilation", 0.f, 0, {0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 1, 0}, {"conv3", 4, 64, 128, 1, 0}, {"conv4", 4, 128, 128, 1, 0}, {"conv5", 3, 128, 256, 1, 0},
      {"conv6", 3, 256, 256, 1, 0}}, {-1, -1, -1}},
    {"dilated_icpr_rate1", 0.f, 0, {0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 1, 0}, {"conv3", 4, 64, 128, 1, 0}, {"conv4", 4, 128, 128, 1, 0}, {"conv5", 3, 128, 256, 1, 0},
      {"conv6", 3, 256, 256, 1, 0}}, {-1, -1, -1}},
    {"dilated_icpr_vary_rate", 0.f, 0, {0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 128, 4, 0}, {"conv4", 4, 128, 128, 1, 0}, {"conv5", 3, 128, 256, 2, 0},
      {"conv6", 3, 256, 256, 4, 0}}, {-1, -1, -1}},
    {"dilated_icpr_old", 0.f, 0, {0}, CHAIN, 256, 3,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv3", 4, 64, 128, 2, 0}, {"conv5", 3, 128, 256, 4, 0}}, {-1, -1, -1}},
    {"dilated_icpr_rate6_avgpool", 0.f, 0, {5, 5, 5, 7, 7, 0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 128, 3, 0}, {"conv4", 4, 128, 128, 4, 0}, {"conv5", 3, 128, 256, 5, 0},
      {"conv6", 3, 256, 256, 6, 0}}, {-1, -1, -1}},
    {"dilated_icpr_rate6_squeeze", 0.f, 0, {0}, SQUEEZE, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 32}, {"conv3", 4, 64, 128, 3, 64}, {"conv4", 4, 128, 128, 4, 64}, {"conv5", 3, 128, 256, 5, 64},
      {"conv6", 3, 256, 256, 6, 128}}, {-1, -1, -1}},
    {"dilated_icpr_rate6_SE", 0.f, 0, {0}, CHAIN, 256, 6,
     {{"conv1", 5, -1, 64, 1, 0}, {"conv2", 5, 64, 64, 2, 0}, {"conv3", 4, 64, 128, 3, 0}, {"conv4", 4, 128, 128, 4, 0}, {"conv5", 3, 128, 256, 5, 0},
      {"conv6", 3, 256, 256, 6, 0}}, {1, 3, 5}},
    {"dilated_icpr_rate6_densely", 0.f, 0, {0}, DENSE, 448, 6,
     {{"conv1", 5, -1, 32, 1, 0}, {"conv2", 5, 32, 32, 2, 0}, {"conv3", 4, 64, 64, 3, 0}, {"conv4", 4, 128, 64, 4, 0}, {"conv5", 3, 192, 128, 5, 0},
      {"conv6", 3, 320, 128, 6, 0}}, {-1, -1, -1}},
};
// isprs:1672 spells Dilated8Pooling 'dilated8_grsl'; contest's 'dilated_grsl_old' (contest:604-641) is layer for layer dilated_grsl
const char* const kAliases[][2] = {{"dilated8_grsl", "dilated_grsl_rate8"}, {"dilated_grsl_old", "dilated_grsl"}};
constexpr int SE_RATIO = 4;
constexpr double BN_DECAY = 0.999;        // tf.contrib.layers.batch_norm default (isprs:658)
constexpr float MOMENTUM = 0.9f;          // isprs:1687
constexpr long long LR_DECAY_STEPS = 50000;   // isprs:1686

const NetTable* find_table(const char* net_type) {
  std::string n = net_type ? net_type : "";
  for (auto& a : kAliases)
    if (n == a[0]) n = a[1];
  for (auto& t : kTables)
    if (n == t.net_type) return &t;
  return nullptr;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------------ plan
struct Layer {
  std::string name;
  int k, cin, cin_k, cout, rate, pad_b, pad_a, halo;
  int src, dst, dst_coff;      // activation slabs: reads channels [0, cin) of src, writes [dst_coff, dst_coff + cout) of dst
  int pool;                    // 0 none, 1 max 3x3, 2 average
  int avg_k;
  int se;                      // index into drs_net::se or -1
  size_t w_off, b_off, bn_off;
};
struct Slab { std::string name; int C, P; };
struct SeBlock { std::string scope; int layer, C, R; size_t w1, b1, w2, b2; };
struct Var { std::string name; size_t off, n; int nd, shape[4]; int where; };   // where: 0 params (grads / momentum alike), 1 bn
struct Buf { std::string name; size_t bytes; int dtype; void* ptr; };           // dtype: 0 f32, 1 f64, 2 u8, 3 i32
enum { F32 = 0, F64 = 1, U8 = 2, I32 = 3 };

// (the three all-reduce kinds: `work` = bytes; timed on the stream the sum is issued on, so that ONE multi-GPU run says where its
//  step went -- 16 latency-bound sync-BN sums, the gradient buffer, the loss / confusion-matrix scalars)
enum Kind { K_CONV_FWD, K_CONV_DGRAD, K_CONV_WGRAD, K_BN_FWD, K_BN_BWD_REDUCE, K_BN_BWD_APPLY, K_CLS, K_MOMENTUM, K_SE_FWD, K_SE_BWD,
            K_AVG_FWD, K_AVG_BWD, K_AR_SYNCBN, K_AR_GRAD, K_AR_SCALARS, K_NKIND };
const char* const kKindNames[K_NKIND] = {"conv_fwd", "conv_dgrad", "conv_wgrad", "bn_act_pool_fwd", "bn_bwd_reduce", "bn_bwd_apply",
                                         "classifier_loss", "momentum_update", "se_fwd", "se_bwd", "avg_pool_fwd", "avg_pool_bwd",
                                         "allreduce_syncbn", "allreduce_grad", "allreduce_scalars"};
struct TimeRec { int kind; double work; hipEvent_t e0, e1; };

// Device pointers of the bound buffers, looked up by name ONCE (resolve(), after the last drs_net_bind) -- a step is ~130 launches
// of a few microseconds of host time each at the per-rank sizes, so its enqueue path does no string building and no map lookup.
struct LayerPtrs {
  float* gzL = nullptr; float* slabL = nullptr;                   // (development library, drs_debug_wgrad_schedule: this layer's own gz / split slab)
  float* z; float* mean_rstd; unsigned char* idx; float* wt;      // raw conv output, (mean, rstd), pool arg-max codes, flipped filter
  const float* w;                                                 // the filter the forward pass multiplies by (conv1: the padded copy)
  float* act_in; float* act_out; float* gact_in; float* gact_out; // slabs this block reads / writes and their gradients
  float *se_act, *se_s, *se_e1, *se_e2;                           // squeeze-and-excitation state of the block (or null)
};
struct Ptrs {
  bool ok = false;
  float *params, *grads, *momentum, *bn, *partial, *gxh, *gz[2], *slab, *w0pad, *conv_ws, *bwd_means, *act, *gpool, *se_scratch;
  float *dw_partial, *db_partial, *logits, *feat_act, *feat_gact;
  size_t conv_ws_floats;
  double *sums, *scalars, *colsum_scratch, *loss_partial, *l2_scratch;
  unsigned char *pred, *labels, *acc_mask, *loss_mask;
  unsigned int* conf;
  std::vector<LayerPtrs> L;
};

}  // namespace

struct drs_net {
  const NetTable* table;
  int channels, K, c_last, b_max, s_max, bessel;
  float alpha, wd, lr_decay;
  long long global_step;
  std::vector<Layer> layers;
  std::vector<Slab> slabs;
  std::vector<SeBlock> se;
  int feat;
  size_t n_params, n_decay, n_bn, cls_w, cls_b, first_bias;
  std::vector<Var> vars;
  std::vector<Buf> bufs;
  std::map<std::string, int> bufidx;
  // data parallelism
  int world, rank;
  drs_allreduce_fn allreduce;
  drs_wait_fn wait;
  void* comm_user;
  // library-side collectives (drs_net_set_rccl, rccl_comm.hip): `rccl_small` carries the latency-bound sums (sync-BN statistics,
  // the loss, the confusion matrix) -- in the forward pass on the compute stream itself (nothing to overlap, no event hand-over),
  // in the backward pass on `small_stream` under the filter gradient of the block above; `rccl_big` carries the gradient
  // buckets on `comm_stream`.  Two communicators, so that a 2 KB sum never queues behind a 4 MB bucket.
  void* rccl_small;
  void* rccl_big;
  hipStream_t comm_stream, small_stream;
  bool own_comm_stream;
  bool rccl_inline;                         // (default) one communicator, every sum on the compute stream itself in program order; DRS_RCCL_ASYNC=1: side streams
  bool rccl_buckets;                        // inline form + the gradient buffer as TWO all-reduces on comm_big's stream, the first under the rest of the backward pass (DRS_RCCL_BUCKETS=2)
  hipEvent_t ev_bucket[3];                  // the three hand-overs of that form
  std::vector<hipEvent_t> comm_events;      // ring: [2 h] = data ready on the compute stream, [2 h + 1] = sum done on the side stream
  int comm_next, comm_ring, comm_in_flight; // ring size (from the number of blocks), asynchronous sums issued since the last wait_handles
  // backward pass of small steps: the filter gradients on a stream of their own beside the batch-norm-backward -> input-gradient
  // chain (two alternating gz slabs); created at first use
  hipStream_t wg_stream;
  hipEvent_t ev_gz[2], ev_wg[2];
  hipEvent_t ev_cls = nullptr;              // two-stream pass: compute stream -> filter-gradient stream (the step has begun; the classifier launch is done)
  hipEvent_t ev_prep = nullptr;             // ... and back: the step's preparation launch (filter flips, zero fills) is done
  int two_stream_mode;                      // -1 by the rule in train_step_impl, 0 never, 1 always (drs_net_set_two_streams)
  // per-slab (B, S) of the pooling call that last zeroed its halo (the halo of a slab one block owns stays zero)
  std::vector<long long> halo_ok;
  bool timing;
  Ptrs ptrs;
#ifdef DRS_DEV
  bool layer_slabs = false;                 // the net lists a gz slab and a split slab per layer (drs_debug_wgrad_schedule before drs_net_create)
  hipStream_t wg_streams[4] = {nullptr, nullptr, nullptr, nullptr};      // [0] = wg_stream
  hipEvent_t ev_gzL[16] = {}, ev_wgS[4] = {}, ev_chain_end = nullptr;
#endif
  std::vector<TimeRec> recs;
  std::vector<hipEvent_t> pool_events;

  Buf* buf(const std::string& n) {
    auto it = bufidx.find(n);
    return it == bufidx.end() ? nullptr : &bufs[it->second];
  }
  template <typename T> T* p(const std::string& n) { Buf* b = buf(n); return b ? static_cast<T*>(b->ptr) : nullptr; }
  void add_buf(const std::string& n, size_t elems, int dtype) {
    static const size_t esz[4] = {4, 8, 1, 4};
    bufidx[n] = (int)bufs.size();
    bufs.push_back(Buf{n, (elems ? elems : 1) * esz[dtype], dtype, nullptr});
  }
  bool is_max(int i) const { return layers[i].pool == 1; }
};

namespace {

void same_pad(int k, int rate, int& pb, int& pa) { const int total = (k - 1) * rate; pb = total / 2; pa = total - pb; }

int slab_index(drs_net* n, const std::string& name, int C) {
  for (size_t i = 0; i < n->slabs.size(); ++i)
    if (n->slabs[i].name == name) return (int)i;
  n->slabs.push_back(Slab{name, C, 0});
  return (int)n->slabs.size() - 1;
}

void add_var(drs_net* n, const std::string& name, size_t off, std::initializer_list<int> shape, int where) {
  Var v;
  v.name = name; v.off = off; v.where = where; v.nd = (int)shape.size(); v.n = 1;
  int i = 0;
  for (int s : shape) { v.shape[i++] = s; v.n *= (size_t)s; }
  for (; i < 4; ++i) v.shape[i] = 1;
  n->vars.push_back(v);
}

// mirrors nets.Plan (first_cin_pad = 8: the exact-fp32 kernels' packed conv1)
void build_plan(drs_net* n) {
  const NetTable& t = *n->table;
  n->alpha = t.alpha;
  n->c_last = t.c_last;
  const int x0 = slab_index(n, "x0", round_up(n->channels, n->channels <= 8 ? 8 : 32));
  struct Blk { std::string name; int k, cin, cout, rate, src, dst, coff; };
  std::vector<Blk> blocks;
  if (t.topo == SQUEEZE) {
    const ConvRow& c0 = t.convs[0];
    int prev = slab_index(n, "c1", c0.cout);
    blocks.push_back({c0.name, c0.k, n->channels, c0.cout, c0.rate, x0, prev, 0});
    for (int j = 1; j < t.nconv; ++j) {
      const ConvRow& c = t.convs[j];
      const int a = slab_index(n, "a" + std::to_string(j + 1), c.kdim), cc = slab_index(n, "c" + std::to_string(j + 1), c.cout);
      blocks.push_back({std::string(c.name) + "_s1", 1, c.cin, c.kdim, c.rate, prev, a, 0});
      blocks.push_back({std::string(c.name) + "_s2_1", 1, c.kdim, c.cout / 2, c.rate, a, cc, 0});
      blocks.push_back({std::string(c.name) + "_s2_2", c.k, c.kdim, c.cout / 2, c.rate, a, cc, c.cout / 2});
      prev = cc;
    }
    n->feat = prev;
  } else if (t.topo == DENSE) {
    const int cat = slab_index(n, "concat", t.c_last);
    int off = 0;
    for (int i = 0; i < t.nconv; ++i) {
      const ConvRow& c = t.convs[i];
      blocks.push_back({c.name, c.k, c.cin < 0 ? n->channels : c.cin, c.cout, c.rate, i == 0 ? x0 : cat, cat, off});
      off += c.cout;
    }
    n->feat = cat;
  } else {
    int prev = x0;
    for (int i = 0; i < t.nconv; ++i) {
      const ConvRow& c = t.convs[i];
      const int dst = slab_index(n, i + 1 < t.nconv ? "x" + std::to_string(i + 1) : std::string("feat"), c.cout);
      blocks.push_back({c.name, c.k, c.cin < 0 ? n->channels : c.cin, c.cout, c.rate, prev, dst, 0});
      prev = dst;
    }
    n->feat = prev;
  }
  bool any_avg = false;
  for (int i = 0; i < 8; ++i) any_avg = any_avg || t.avg_pool[i] > 0;
  for (size_t i = 0; i < blocks.size(); ++i) {
    const Blk& b = blocks[i];
    Layer L;
    L.name = b.name; L.k = b.k; L.cin = b.cin; L.cout = b.cout; L.rate = b.rate;
    same_pad(b.k, b.rate, L.pad_b, L.pad_a);
    L.halo = std::max(L.pad_b, L.pad_a);
    n->slabs[b.src].P = std::max(n->slabs[b.src].P, L.halo);      // a slab's halo must cover every conv that reads it
    L.cin_k = b.src == x0 ? n->slabs[x0].C : round_up(b.cin, 32);
    L.src = b.src; L.dst = b.dst; L.dst_coff = b.coff;
    L.pool = t.max_pool ? 1 : (any_avg && i < 8 && t.avg_pool[i] > 0 ? 2 : 0);
    L.avg_k = L.pool == 2 ? t.avg_pool[i] : 0;
    L.se = -1;
    n->layers.push_back(L);
  }
  // flat parameter layout: every kernel (HWIO), then every bias; the classifier last in both groups
  size_t off = 0;
  for (auto& L : n->layers) {
    L.w_off = off;
    add_var(n, L.name + "/weights", off, {L.k, L.k, L.cin, L.cout}, 0);
    off += (size_t)L.k * L.k * L.cin * L.cout;
  }
  n->cls_w = off;
  add_var(n, "conv_classifier/weights", off, {1, 1, n->c_last, n->K}, 0);
  off += (size_t)n->c_last * n->K;
  for (int j = 0; j < 3; ++j) {
    const int li = t.se_after[j];
    if (li < 0) continue;
    SeBlock s;
    s.scope = "se" + std::to_string(j + 1); s.layer = li; s.C = n->layers[li].cout; s.R = s.C / SE_RATIO;
    s.w1 = off; add_var(n, s.scope + "_fc1/weights", off, {s.C, s.R}, 0); off += (size_t)s.C * s.R;
    s.w2 = off; add_var(n, s.scope + "_fc2/weights", off, {s.R, s.C}, 0); off += (size_t)s.R * s.C;
    n->layers[li].se = (int)n->se.size();
    n->se.push_back(s);
  }
  n->n_decay = off;                                   // weight decay applies to kernels only (isprs:640-652)
  n->first_bias = off;
  for (auto& L : n->layers) {
    L.b_off = off;
    add_var(n, L.name + "/biases", off, {L.cout}, 0);
    off += L.cout;
  }
  n->cls_b = off;
  add_var(n, "conv_classifier/biases", off, {n->K}, 0);
  off += n->K;
  for (auto& s : n->se) {
    s.b1 = off; add_var(n, s.scope + "_fc1/biases", off, {s.R}, 0); off += s.R;
    s.b2 = off; add_var(n, s.scope + "_fc2/biases", off, {s.C}, 0); off += s.C;
  }
  n->n_params = off;
  off = 0;
  for (auto& L : n->layers) {
    L.bn_off = off;
    add_var(n, L.name + "/moving_mean", off, {L.cout}, 1);
    add_var(n, L.name + "/moving_variance", off + L.cout, {L.cout}, 1);
    off += 2 * (size_t)L.cout;
  }
  n->n_bn = off;
}

#ifdef DRS_DEV
int g_wg_layer_slabs = 0;      // drs_debug_wgrad_schedule (see train_step_impl)
#endif

// every device buffer a step touches, sized once for (b_max, s_max); the caller allocates and binds them
void list_buffers(drs_net* n) {
  const size_t B = n->b_max, S = n->s_max, M = B * S * S;
  n->add_buf("params", n->n_params, F32);
  n->add_buf("grads", n->n_params, F32);
  n->add_buf("momentum", n->n_params, F32);
  n->add_buf("bn", n->n_bn, F32);
  for (size_t i = 0; i < n->slabs.size(); ++i) {
    const Slab& s = n->slabs[i];
    n->add_buf("act:" + s.name, B * (S + 2 * s.P) * (S + 2 * s.P) * s.C, F32);     // zero-haloed activation slab
    if (i != 0) n->add_buf("gact:" + s.name, M * s.C, F32);                       // gradient wrt it, [M][C]
  }
  int cmax = 0, hmax = 0;
  size_t rows_fwd = 0, part = 0, slab = 0, conv_ws = 0;
  for (size_t i = 0; i < n->layers.size(); ++i) {
    const Layer& L = n->layers[i];
    const std::string id = std::to_string(i);
    n->add_buf("z" + id, M * L.cout, F32);
    if (L.pool == 1) n->add_buf("idx" + id, M * L.cout, U8);
    n->add_buf("mean_rstd" + id, 2 * (size_t)L.cout, F32);
    cmax = std::max(cmax, L.cout);
    hmax = std::max(hmax, L.halo);
    const size_t mt = drs_conv_mtile(L.cout);
    rows_fwd = std::max(rows_fwd, (M + mt - 1) / mt);
    // the backward slab's row count is not monotonic in the batch or the patch side: size it over every (b, s) a step may use
    for (int b = 1; b <= n->b_max; ++b)
      for (int s = 1; s <= n->s_max; ++s)
        part = std::max(part, (size_t)drs_bn_backward_rows(b, s, L.cout, L.pool == 1) * L.cout * 2);
    slab = std::max(slab, (size_t)drs_conv_wgrad_splits(n->b_max, n->s_max, L.k, L.cin_k, L.cout) * L.k * L.k * L.cin_k * L.cout);
    if (i > 0) n->add_buf("wt" + id, (size_t)L.k * L.k * L.cin * L.cout, F32);   // flipped / transposed filter of the input-gradient pass
    conv_ws = std::max(conv_ws, std::max(drs_conv_workspace_floats(L.cout), i > 0 ? drs_conv_workspace_floats(L.cin) : (size_t)0));
  }
  n->add_buf("conv_ws", conv_ws, F32);          // partial-sum slab of the stream-K convolution launches (forward: N = cout, input gradient: N = cin)
  n->add_buf("sums", 2 * (size_t)cmax, F64);
  n->add_buf("bwd_means", 2 * (size_t)cmax, F32);      // (sum g, sum g xhat) / count as fp32: drs_bn_backward_apply_means' table (single rank)
  n->add_buf("colsum_scratch", drs_colsum_scratch_doubles(std::max(2 * cmax, n->c_last * n->K)), F64);
  n->add_buf("partial", std::max(rows_fwd * cmax * 2, part), F32);
  n->add_buf("gxh", M * cmax, F32);
  n->add_buf("gz", B * (S + 2 * hmax) * (S + 2 * hmax) * cmax, F32);
  n->add_buf("gz2", B * (S + 2 * hmax) * (S + 2 * hmax) * cmax, F32);       // the second haloed output-gradient slab of the two-stream backward pass
  n->add_buf("slab", slab, F32);
#ifdef DRS_DEV
  n->layer_slabs = g_wg_layer_slabs != 0 && n->layers.size() <= 16;
  if (n->layer_slabs)
    for (size_t i = 0; i < n->layers.size(); ++i) {
      const Layer& L = n->layers[i];
      n->add_buf("gzL" + std::to_string(i), B * (S + 2 * L.halo) * (S + 2 * L.halo) * L.cout, F32);
      n->add_buf("slabL" + std::to_string(i), (size_t)drs_conv_wgrad_splits(n->b_max, n->s_max, L.k, L.cin_k, L.cout) * L.k * L.k * L.cin_k * L.cout, F32);
    }
#endif
  const Layer& L0 = n->layers[0];
  n->add_buf("w0pad", (size_t)round_up(L0.k * L0.k * L0.cin_k, 32) * L0.cout, F32);
  bool any_avg = false;
  for (auto& L : n->layers) any_avg = any_avg || L.pool == 2;
  if (!n->se.empty() || any_avg) {
    n->add_buf("act", M * cmax, F32);        // activated, not yet averaged / scaled output of a block
    n->add_buf("gpool", M * cmax, F32);      // gradient wrt it
  }
  for (size_t j = 0; j < n->se.size(); ++j) {
    const SeBlock& s = n->se[j];
    const std::string id = std::to_string(j);
    n->add_buf("se_act" + id, M * s.C, F32);
    n->add_buf("se_s" + id, B * s.C, F32);
    n->add_buf("se_e1" + id, B * s.R, F32);
    n->add_buf("se_e2" + id, B * s.C, F32);
  }
  if (!n->se.empty()) n->add_buf("se_scratch", B * (3 * (size_t)cmax + cmax / 4), F32);
  const size_t crow = drs_classifier_rows(n->b_max, n->s_max);
  n->add_buf("dw_partial", crow * n->c_last * n->K, F32);
  n->add_buf("db_partial", crow * n->K, F32);
  n->add_buf("loss_partial", crow, F64);
  n->add_buf("scalars", 4, F64);             // [0] mean CE (all ranks), [1] 0.5 * sum w^2
  n->add_buf("l2_scratch", 256, F64);
  n->add_buf("logits", M * n->K, F32);
  n->add_buf("pred", M, U8);
  n->add_buf("conf", (size_t)n->K * n->K, I32);
  n->add_buf("labels", M, U8);
  n->add_buf("acc_mask", M, U8);
  n->add_buf("loss_mask", M, U8);
}

// HIP-event bracket of one family's launches (drs_net_timing).  Events are recycled through drs_net::pool_events, and a run that
// never asks for a summary stops recording at MAX_RECS instead of growing without bound.
constexpr size_t MAX_RECS = 1 << 16;
struct Timed {
  drs_net* n; hipStream_t st; int kind; TimeRec rec; bool on;
  static bool take(drs_net* n, hipEvent_t* e) {
    if (!n->pool_events.empty()) { *e = n->pool_events.back(); n->pool_events.pop_back(); return true; }
    return hipEventCreate(e) == hipSuccess;
  }
  Timed(drs_net* n_, hipStream_t st_, int kind_, double work) : n(n_), st(st_), kind(kind_), on(n_->timing && n_->recs.size() < MAX_RECS) {
    if (!on) return;
    rec.kind = kind; rec.work = work;
    if (!take(n, &rec.e0)) { on = false; return; }
    if (!take(n, &rec.e1)) { n->pool_events.push_back(rec.e0); on = false; return; }
    (void)hipEventRecord(rec.e0, st);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(rec.e1, st);
    n->recs.push_back(rec);
  }
};

#define DRS_TRY(expr) do { const int rc_ = (expr); if (rc_ != DRS_OK) return rc_; } while (0)

#ifdef DRS_DEV
// Schedule fuzzing (development library, drs_debug_jitter): a launch that sleeps, put on one of the step's streams at the points where
// they hand work to each other -- up to 150 us, one time in eight up to 4 ms (the slack some hand-overs have: the step's preparation
// launch is needed a whole forward pass after it was issued).  It moves every cross-stream dependency of the two-stream pass off its
// usual timing: a missing event wait that the usual timing hides becomes a different result
// (tests/test_gpu_determinism.py::test_sleeps_on_the_steps_streams_do_not_change_a_bit holds the jittered steps to the unjittered bits).
__global__ void jitter_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
unsigned long long g_jitter_state = 0;      // 0 = off
int g_wg_stream_prio = 2;                   // drs_debug_wg_stream_prio: the filter-gradient stream's priority arm (2 = the product's: highest)
// r06 experiment (drs_debug_wgrad_schedule): the filter gradients of a two-stream step with a gz slab and a split
// slab PER LAYER (set before drs_net_create: the net then lists "gzL<i>" / "slabL<i>"), so that the chain never waits for a filter
// gradient to release a slab, on `streams` filter-gradient streams (layer i on stream i % streams: their launches overlap each other's
// fill and drain -- what ONE grouped launch over the layers would do), issued as the chain goes (defer 0), or all after the chain's
// last launch, each waiting only for its own gz (defer 2) or for the END of the chain (defer 1: the chain runs uncontended).
int g_wg_streams = 1, g_wg_defer = 0;       // (g_wg_layer_slabs is declared above list_buffers, which reads it)
int g_reductions_on_chain = 0;              // drs_debug_reductions_on_chain: the classifier's slab reductions stay on the compute stream
inline void jitter(hipStream_t s) {
  if (!g_jitter_state) return;
  g_jitter_state = g_jitter_state * 6364136223846793005ull + 1442695040888963407ull;
  const unsigned r = (unsigned)(g_jitter_state >> 36);
  if (!(r & 1)) return;
  const unsigned span = ((r >> 1) & 7) == 0 ? 400000u : 15000u;      // 100 MHz ticks
  hipLaunchKernelGGL(jitter_kernel, dim3(1), dim3(64), 0, s, (long long)((r >> 4) % span));
}
#define DRS_JITTER(s) jitter(s)
#else
#define DRS_JITTER(s) do { } while (0)
#endif

// a non-blocking stream at a stream priority of its own (see train_step_impl: a hardware-queue pool apart from the caller's); a runtime
// that refuses the priority gets a plain stream -- the placement is an optimisation, never a reason to fail a step
inline hipError_t stream_at_priority(hipStream_t* s, int prio) {
  (void)hipGetLastError();
  if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio) == hipSuccess) return hipSuccess;
  (void)hipGetLastError();
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// every sum over ranks goes through the host's callback; with a callback installed it does so at world == 1 too (an identity there:
// lets a one-GPU box drive the whole collective path, RCCL included)
inline bool collectives(const drs_net* n) { return n->world > 1 || n->allreduce != nullptr || n->rccl_small != nullptr; }

constexpr size_t SMALL_BYTES = 16384;     // sums up to here go to the small communicator
constexpr size_t MAX_SLABS = 32;          // activation slabs of a net (the deepest table has 13); checked in drs_net_create

// `big_only`: an asynchronous sum that must not touch the small communicator whatever its size (two-stream backward pass: the
// small communicator is then driven from the compute stream alone, so every asynchronous sum goes to the big one's stream --
// one communicator is never driven from two streams that no event orders)
// Timed on the stream the sum RUNS on, around the collective call itself: `st` for the inline and the synchronous forms, the side
// stream behind its wait for the data for an asynchronous library-side sum.  A sum that goes through the host's callback runs on a
// stream of the host's that this side cannot bracket: no record then (bench.py reports no allreduce_* row rather than the time of
// an event hand-over).
int all_reduce(drs_net* n, int kind, void* ptr, size_t count, int dtype, int async, hipStream_t st, int* handle, bool big_only = false) {
  static const size_t esz[4] = {4, 8, 1, 4};
  if (handle) *handle = -1;
  if (!collectives(n)) return DRS_OK;
  const double bytes = (double)(count * esz[dtype]);
  if (n->rccl_small) {
    if (n->rccl_inline) { Timed t(n, st, kind, bytes); return drs_rccl_all_reduce_sum(n->rccl_small, ptr, count, dtype, st); }
    const bool small = !n->rccl_big || (!(big_only && async) && count * esz[dtype] <= SMALL_BYTES);
    if (!async) { Timed t(n, st, kind, bytes); return drs_rccl_all_reduce_sum(small ? n->rccl_small : n->rccl_big, ptr, count, dtype, st); }
    if (n->comm_in_flight >= n->comm_ring) return DRS_ERR_ARG;      // an event slot still waited for would be recorded again
    ++n->comm_in_flight;
    const int h = n->comm_next;
    n->comm_next = (h + 1) % n->comm_ring;
    hipStream_t side = small ? n->small_stream : n->comm_stream;
    if (hipEventRecord(n->comm_events[2 * h], st) != hipSuccess) return DRS_ERR_HIP;
    if (hipStreamWaitEvent(side, n->comm_events[2 * h], 0) != hipSuccess) return DRS_ERR_HIP;
    {
      Timed t(n, side, kind, bytes);
      DRS_TRY(drs_rccl_all_reduce_sum(small ? n->rccl_small : n->rccl_big, ptr, count, dtype, side));
    }
    if (hipEventRecord(n->comm_events[2 * h + 1], side) != hipSuccess) return DRS_ERR_HIP;
    if (handle) *handle = h;
    return DRS_OK;
  }
  if (!n->allreduce) return DRS_ERR_ARG;
  const int h = n->allreduce(n->comm_user, ptr, count, dtype, async, st);
  if (h < 0) return DRS_ERR_HIP;
  if (handle) *handle = h;
  return DRS_OK;
}

int wait_handles(drs_net* n, const int* hs, size_t count, hipStream_t st) {
  if (n->rccl_small) {
    for (size_t i = 0; i < count; ++i) {
      const int h = hs[i];
      if (h < 0) continue;
      if (hipStreamWaitEvent(st, n->comm_events[2 * h + 1], 0) != hipSuccess) return DRS_ERR_HIP;
      --n->comm_in_flight;
    }
    return DRS_OK;
  }
  if (!collectives(n) || !n->wait) return DRS_OK;
  for (size_t i = 0; i < count; ++i)
    if (hs[i] >= 0 && n->wait(n->comm_user, hs[i], st) != 0) return DRS_ERR_HIP;
  return DRS_OK;
}

// every buffer bound -> the pointer table of the step (false: something is still unbound)
bool resolve(drs_net* n) {
  Ptrs& P = n->ptrs;
  if (P.ok) return true;
  for (auto& b : n->bufs)
    if (!b.ptr) return false;
  P.params = n->p<float>("params"); P.grads = n->p<float>("grads"); P.momentum = n->p<float>("momentum"); P.bn = n->p<float>("bn");
  P.partial = n->p<float>("partial"); P.gxh = n->p<float>("gxh"); P.gz[0] = n->p<float>("gz"); P.gz[1] = n->p<float>("gz2");
  P.slab = n->p<float>("slab"); P.w0pad = n->p<float>("w0pad"); P.conv_ws = n->p<float>("conv_ws");
  P.conv_ws_floats = n->buf("conv_ws")->bytes / sizeof(float);
  P.bwd_means = n->p<float>("bwd_means"); P.act = n->p<float>("act"); P.gpool = n->p<float>("gpool"); P.se_scratch = n->p<float>("se_scratch");
  P.dw_partial = n->p<float>("dw_partial"); P.db_partial = n->p<float>("db_partial"); P.logits = n->p<float>("logits");
  P.sums = n->p<double>("sums"); P.scalars = n->p<double>("scalars"); P.colsum_scratch = n->p<double>("colsum_scratch");
  P.loss_partial = n->p<double>("loss_partial"); P.l2_scratch = n->p<double>("l2_scratch");
  P.pred = n->p<unsigned char>("pred"); P.labels = n->p<unsigned char>("labels"); P.acc_mask = n->p<unsigned char>("acc_mask");
  P.loss_mask = n->p<unsigned char>("loss_mask"); P.conf = n->p<unsigned int>("conf");
  const Slab& f = n->slabs[n->feat];
  P.feat_act = n->p<float>("act:" + f.name); P.feat_gact = n->p<float>("gact:" + f.name);
  P.L.assign(n->layers.size(), LayerPtrs());
  for (size_t i = 0; i < n->layers.size(); ++i) {
    const Layer& L = n->layers[i];
    const std::string id = std::to_string(i);
    LayerPtrs& q = P.L[i];
    q.z = n->p<float>("z" + id); q.mean_rstd = n->p<float>("mean_rstd" + id); q.idx = n->p<unsigned char>("idx" + id);
    q.wt = i > 0 ? n->p<float>("wt" + id) : nullptr;
    q.w = i == 0 ? P.w0pad : P.params + L.w_off;
    q.act_in = n->p<float>("act:" + n->slabs[L.src].name); q.act_out = n->p<float>("act:" + n->slabs[L.dst].name);
    q.gact_in = L.src != 0 ? n->p<float>("gact:" + n->slabs[L.src].name) : nullptr;
    q.gact_out = n->p<float>("gact:" + n->slabs[L.dst].name);
    q.se_act = q.se_s = q.se_e1 = q.se_e2 = nullptr;
#ifdef DRS_DEV
    if (n->layer_slabs) { q.gzL = n->p<float>("gzL" + id); q.slabL = n->p<float>("slabL" + id); }
#endif
    if (L.se >= 0) {
      const std::string sid = std::to_string(L.se);
      q.se_act = n->p<float>("se_act" + sid); q.se_s = n->p<float>("se_s" + sid); q.se_e1 = n->p<float>("se_e1" + sid); q.se_e2 = n->p<float>("se_e2" + sid);
    }
  }
  P.ok = true;
  return true;
}

// conv -> (+bias) -> batch norm -> activation -> pool / SE, for every block (net.py _forward_layers, exact-fp32 arithmetic)
int forward_layers(drs_net* n, int B, int S, bool training, double count, hipStream_t st) {
  const long long M = (long long)B * S * S;
  const Ptrs& P = n->ptrs;
  float* params = P.params;
  float* bn = P.bn;
  const Layer& L0 = n->layers[0];
  DRS_TRY(drs_filter_pad_cin(params + L0.w_off, P.w0pad, L0.k, L0.cin, L0.cin_k, L0.cout, st));
  float* partial = P.partial;
  double* sums = P.sums;
  for (size_t i = 0; i < n->layers.size(); ++i) {
    const Layer& L = n->layers[i];
    const LayerPtrs& q = P.L[i];
    const Slab& in = n->slabs[L.src];
    float* z = q.z;
    float* mr = q.mean_rstd;
    {
      Timed t(n, st, K_CONV_FWD, 2.0 * M * L.k * L.k * L.cin * L.cout);
      DRS_TRY(drs_conv_forward_ws(q.act_in, B, S, in.P, in.C, 0, q.w, params + L.b_off, L.k, L.rate, L.pad_b,
                                  L.cin_k, L.cout, z, L.cout, 0, 0, training ? partial : nullptr, P.conv_ws, P.conv_ws_floats, st));
    }
    float* mm = bn + L.bn_off;
    float* mv = mm + L.cout;
    bool fold_finish = false;
    if (training && !collectives(n)) {
      DRS_TRY(drs_conv_stats_finish(partial, (int)M, drs_conv_mtile(L.cout), L.cout, count, mr, mm, mv, BN_DECAY, n->bessel, nullptr, st));
    } else if (training) {
      DRS_TRY(drs_conv_stats_reduce(partial, (int)M, drs_conv_mtile(L.cout), L.cout, sums, nullptr, st));
      DRS_TRY(all_reduce(n, K_AR_SYNCBN, sums, 2 * (size_t)L.cout, F64, 0, st, nullptr));          // sync batch norm over the global batch
      // (a plain pooled block: the kernel that normalises works mean / rstd / moving averages out of the sums itself, below)
      fold_finish = L.pool == 1 && L.se < 0 && L.cout <= 512;
      if (!fold_finish) DRS_TRY(drs_bn_finish(sums, count, L.cout, mr, mm, mv, BN_DECAY, n->bessel, st));
    } else {
      DRS_TRY(drs_bn_eval_coeffs(mm, mv, L.cout, mr, st));
    }
    const Slab& out = n->slabs[L.dst];
    float* outp = q.act_out;
    const bool mx = L.pool == 1;
    unsigned char* idx = (training && mx) ? q.idx : nullptr;
    if (L.se >= 0) {      // activation into a plain [M][C] buffer, then squeeze-and-excitation scaling into the next slab
      const SeBlock& s = n->se[L.se];
      float* act = q.se_act;
      { Timed t(n, st, K_BN_FWD, M * L.cout * 8.0);
        DRS_TRY(drs_bn_act_pool_forward(z, B, S, L.cout, mr, n->alpha, 0, act, 0, L.cout, 0, nullptr, st)); }
      Timed t(n, st, K_SE_FWD, M * L.cout * 12.0);
      DRS_TRY(drs_se_forward(act, B, S, L.cout, s.R, params + s.w1, params + s.b1, params + s.w2, params + s.b2, q.se_s, q.se_e1, q.se_e2, outp,
                             out.P, out.C, L.dst_coff, st));
      n->halo_ok[L.dst] = -1;
    } else if (L.pool == 2) {   // activation into a plain [M][C] buffer, then the k x k average into the next layer's slab
      float* act = P.act;
      { Timed t(n, st, K_BN_FWD, M * L.cout * 8.0);
        DRS_TRY(drs_bn_act_pool_forward(z, B, S, L.cout, mr, n->alpha, 0, act, 0, L.cout, 0, nullptr, st)); }
      Timed t(n, st, K_AVG_FWD, M * L.cout * 8.0);
      DRS_TRY(drs_avg_pool_forward(act, B, S, L.cout, L.avg_k, outp, out.P, out.C, L.dst_coff, st));
      n->halo_ok[L.dst] = -1;
    } else {
      // the halo of a slab this block owns alone stays zero between calls of the same geometry: do not rewrite it
      const bool whole = out.C == L.cout && L.dst_coff == 0;
      const long long key = ((long long)B << 20) | S;
      const int hz = (mx && whole && n->halo_ok[L.dst] == key) ? 2 : 0;
      n->halo_ok[L.dst] = (mx && whole) ? key : -1;
      Timed t(n, st, K_BN_FWD, M * L.cout * ((training && mx) ? 9.0 : 8.0));
      if (fold_finish)
        DRS_TRY(drs_bn_finish_act_pool_forward(sums, count, mr, mm, mv, BN_DECAY, n->bessel, z, B, S, L.cout, n->alpha, 1 | hz, outp, out.P, out.C,
                                               L.dst_coff, idx, st));
      else
        DRS_TRY(drs_bn_act_pool_forward(z, B, S, L.cout, mr, n->alpha, (mx ? 1 : 0) | hz, outp, out.P, out.C, L.dst_coff, idx, st));
    }
  }
  return DRS_OK;
}

bool check_bs(const drs_net* n, int B, int S) { return B >= 1 && S >= 1 && B <= n->b_max && S <= n->s_max; }

}  // namespace

extern "C" {

#ifdef DRS_DEV
int drs_debug_jitter(unsigned long long seed) { g_jitter_state = seed; return 0; }
int drs_debug_wg_stream_prio(int arm) { const int old = g_wg_stream_prio; if (arm >= 0) g_wg_stream_prio = arm; return old; }
int drs_debug_reductions_on_chain(int on) { const int old = g_reductions_on_chain; if (on >= 0) g_reductions_on_chain = on; return old; }
int drs_debug_wgrad_schedule(int layer_slabs, int streams, int defer) {
  if (layer_slabs >= 0) g_wg_layer_slabs = layer_slabs;      // read by drs_net_create: nets made afterwards list the per-layer slabs
  if (streams >= 1) g_wg_streams = streams;                  // read per step
  if (defer >= 0) g_wg_defer = defer;                        // read per step
  return 0;
}
#endif

int drs_net_create(const char* net_type, int channels, int num_classes, float weight_decay, int b_max, int s_max, int bessel_moving_var,
                   float lr_decay_factor, drs_net_t** out) {
  if (!out) return DRS_ERR_ARG;
  *out = nullptr;
  const NetTable* t = find_table(net_type);
  // the reference prints a red message and carries on with logits = None (isprs:1679-1680); the library rejects the name
  if (!t || channels < 1 || channels > 32 || num_classes < 1 || num_classes > 8 || b_max < 1 || s_max < 1) return DRS_ERR_ARG;
  if ((long long)b_max * s_max * s_max >= (1 << 24)) return DRS_ERR_ARG;
  drs_net* n = nullptr;
  try {                                   // nothing throws across the boundary: an allocation failure of the host-side tables is a status
    n = new drs_net();
    n->table = t; n->channels = channels; n->K = num_classes; n->wd = weight_decay; n->b_max = b_max; n->s_max = s_max;
    n->bessel = bessel_moving_var ? 1 : 0; n->lr_decay = lr_decay_factor; n->global_step = 0;
    n->world = 1; n->rank = 0; n->allreduce = nullptr; n->wait = nullptr; n->comm_user = nullptr; n->timing = false;
    n->rccl_small = n->rccl_big = nullptr; n->comm_stream = n->small_stream = nullptr; n->own_comm_stream = false; n->comm_next = 0;
    n->rccl_inline = false; n->rccl_buckets = false; n->comm_ring = 0; n->comm_in_flight = 0;
    for (auto& e : n->ev_bucket) e = nullptr;
    n->wg_stream = nullptr;
    n->two_stream_mode = -1;
    build_plan(n);
    if (n->slabs.size() > MAX_SLABS) { delete n; return DRS_ERR_ARG; }
    list_buffers(n);
    n->halo_ok.assign(n->slabs.size(), -1);
  } catch (...) {
    delete n;
    return DRS_ERR_HIP;
  }
  *out = n;
  return DRS_OK;
}

static void release_rccl(drs_net* n) {
  for (auto e : n->comm_events) (void)hipEventDestroy(e);
  n->comm_events.clear();
  if (n->small_stream) (void)hipStreamDestroy(n->small_stream);
  if (n->own_comm_stream && n->comm_stream) (void)hipStreamDestroy(n->comm_stream);
  n->small_stream = n->comm_stream = nullptr;
  n->own_comm_stream = false;
  n->rccl_small = n->rccl_big = nullptr;
  n->rccl_inline = false;
  n->rccl_buckets = false;
  for (auto& e : n->ev_bucket) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  n->comm_in_flight = 0;
}

void drs_net_destroy(drs_net_t* n) {
  if (!n) return;
  release_rccl(n);
  if (n->wg_stream) {
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(n->ev_gz[i]); (void)hipEventDestroy(n->ev_wg[i]); }
    if (n->ev_cls) (void)hipEventDestroy(n->ev_cls);
    if (n->ev_prep) (void)hipEventDestroy(n->ev_prep);
    (void)hipStreamDestroy(n->wg_stream);
  }
#ifdef DRS_DEV
  for (int j = 1; j < 4; ++j) if (n->wg_streams[j]) (void)hipStreamDestroy(n->wg_streams[j]);
  for (auto e : n->ev_wgS) if (e) (void)hipEventDestroy(e);
  for (auto e : n->ev_gzL) if (e) (void)hipEventDestroy(e);
  if (n->ev_chain_end) (void)hipEventDestroy(n->ev_chain_end);
#endif
  for (auto& r : n->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : n->pool_events) (void)hipEventDestroy(e);
  delete n;
}

int drs_net_num_buffers(const drs_net_t* n) { return n ? (int)n->bufs.size() : 0; }

int drs_net_buffer_info(const drs_net_t* n, int index, char* name, int name_cap, size_t* bytes, int* dtype) {
  if (!n || index < 0 || index >= (int)n->bufs.size()) return DRS_ERR_ARG;
  const Buf& b = n->bufs[index];
  if (name && name_cap > 0) { std::strncpy(name, b.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (bytes) *bytes = b.bytes;
  if (dtype) *dtype = b.dtype;
  return DRS_OK;
}

int drs_net_bind(drs_net_t* n, const char* name, void* dev_ptr, size_t bytes) {
  if (!n || !name || !dev_ptr) return DRS_ERR_ARG;
  Buf* b = n->buf(name);
  if (!b || bytes < b->bytes) return DRS_ERR_ARG;
  b->ptr = dev_ptr;
  n->ptrs.ok = false;
  std::fill(n->halo_ok.begin(), n->halo_ok.end(), -1);
  return DRS_OK;
}

int drs_net_buffer(drs_net_t* n, const char* name, void** dev_ptr, size_t* bytes) {
  if (!n || !name) return DRS_ERR_ARG;
  Buf* b = n->buf(name);
  if (!b) return DRS_ERR_ARG;
  if (dev_ptr) *dev_ptr = b->ptr;
  if (bytes) *bytes = b->bytes;
  return DRS_OK;
}

int drs_grad_buffer(drs_net_t* n, float** dev_ptr, size_t* count) {
  if (!n) return DRS_ERR_ARG;
  if (dev_ptr) *dev_ptr = n->p<float>("grads");      // (not a per-step call)
  if (count) *count = n->n_params;
  return DRS_OK;
}

int drs_net_num_variables(const drs_net_t* n) { return n ? (int)n->vars.size() : 0; }

int drs_net_variable_info(const drs_net_t* n, int index, char* name, int name_cap, size_t* offset, size_t* count, int* shape4, int* in_bn) {
  if (!n || index < 0 || index >= (int)n->vars.size()) return DRS_ERR_ARG;
  const Var& v = n->vars[index];
  if (name && name_cap > 0) { std::strncpy(name, v.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = v.off;
  if (count) *count = v.n;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = i < v.nd ? v.shape[i] : 0;
  if (in_bn) *in_bn = v.where;
  return DRS_OK;
}

int drs_net_layer_info(const drs_net_t* n, int index, char* name, int name_cap, int* geom8, char* src_slab, char* dst_slab, int slab_cap,
                       int* dst_coff, int* pool) {
  if (!n || index < 0 || index >= (int)n->layers.size()) return DRS_ERR_ARG;
  const Layer& L = n->layers[index];
  auto put = [](char* dst, int cap, const std::string& v) { if (dst && cap > 0) { std::strncpy(dst, v.c_str(), cap - 1); dst[cap - 1] = 0; } };
  put(name, name_cap, L.name);
  put(src_slab, slab_cap, n->slabs[L.src].name);
  put(dst_slab, slab_cap, n->slabs[L.dst].name);
  if (geom8) { const int g[8] = {L.k, L.rate, L.cin, L.cin_k, L.cout, L.pad_b, L.pad_a, L.halo}; for (int i = 0; i < 8; ++i) geom8[i] = g[i]; }
  if (dst_coff) *dst_coff = L.dst_coff;
  if (pool) *pool = L.pool == 2 ? 2 + 256 * L.avg_k : L.pool;
  return DRS_OK;
}

// the net as a whole: the canonical net_type (aliases resolved: 'dilated8_grsl' -> 'dilated_grsl_rate8'), the activation's alpha
// (max(alpha x, x): 0 = ReLU, 0.1 = leaky ReLU, isprs:620-621), the classifier's input width, the slab the classifier reads, the
// topology (0 chain, 1 dense concat isprs:921-948, 2 squeeze isprs:726-742) and the number of squeeze-and-excitation blocks
int drs_net_info(const drs_net_t* n, char* net_type, int name_cap, float* alpha, int* c_last, char* feat_slab, int slab_cap, int* topology,
                 int* n_se) {
  if (!n) return DRS_ERR_ARG;
  auto put = [](char* dst, int cap, const std::string& v) { if (dst && cap > 0) { std::strncpy(dst, v.c_str(), cap - 1); dst[cap - 1] = 0; } };
  put(net_type, name_cap, n->table->net_type);
  put(feat_slab, slab_cap, n->slabs[n->feat].name);
  if (alpha) *alpha = n->alpha;
  if (c_last) *c_last = n->c_last;
  if (topology) *topology = (int)n->table->topo;
  if (n_se) *n_se = (int)n->se.size();
  return DRS_OK;
}

// squeeze-and-excitation block `index` (isprs:682-697, 1042-1050): its scope ("se1": variables <scope>_fc1/weights, ...), the block
// whose activation it scales, channels and the reduced width C / 4
int drs_net_se_info(const drs_net_t* n, int index, char* scope, int scope_cap, int* layer, int* channels, int* reduced) {
  if (!n || index < 0 || index >= (int)n->se.size()) return DRS_ERR_ARG;
  const SeBlock& s = n->se[index];
  if (scope && scope_cap > 0) { std::strncpy(scope, s.scope.c_str(), scope_cap - 1); scope[scope_cap - 1] = 0; }
  if (layer) *layer = s.layer;
  if (channels) *channels = s.C;
  if (reduced) *reduced = s.R;
  return DRS_OK;
}

// the net_type strings drs_net_create accepts (the if-chains isprs:1660-1680, coffee:1188-1215, contest:995-1012), one per index:
// the tables first, then the aliases; *canonical = index of the table an alias resolves to (its own index for a table)
int drs_net_type_name(int index, char* name, int name_cap, int* canonical) {
  const int nt = (int)(sizeof(kTables) / sizeof(kTables[0])), na = (int)(sizeof(kAliases) / sizeof(kAliases[0]));
  if (index < 0 || index >= nt + na) return DRS_ERR_ARG;
  const char* v = index < nt ? kTables[index].net_type : kAliases[index - nt][0];
  if (name && name_cap > 0) { std::strncpy(name, v, name_cap - 1); name[name_cap - 1] = 0; }
  if (canonical) {
    *canonical = index;
    if (index >= nt)
      for (int i = 0; i < nt; ++i)
        if (std::string(kTables[i].net_type) == kAliases[index - nt][1]) *canonical = i;
  }
  return DRS_OK;
}

int drs_net_layout(const drs_net_t* n, size_t* n_params, size_t* n_decay, size_t* n_bn, int* n_layers, int* x0_channels, int* x0_halo) {
  if (!n) return DRS_ERR_ARG;
  if (n_params) *n_params = n->n_params;
  if (n_decay) *n_decay = n->n_decay;
  if (n_bn) *n_bn = n->n_bn;
  if (n_layers) *n_layers = (int)n->layers.size();
  if (x0_channels) *x0_channels = n->slabs[0].C;
  if (x0_halo) *x0_halo = n->slabs[0].P;
  return DRS_OK;
}

static int var_span(drs_net_t* n, const char* name, const char* slot, float** base, size_t* count) {
  if (!n || !name) return DRS_ERR_ARG;
  for (auto& v : n->vars) {
    if (v.name != name) continue;
    const bool mom = slot && std::strcmp(slot, "Momentum") == 0;
    if (mom && v.where != 0) return DRS_ERR_ARG;
    float* b = v.where == 1 ? n->p<float>("bn") : (mom ? n->p<float>("momentum") : n->p<float>("params"));
    if (!b) return DRS_ERR_ARG;
    *base = b + v.off;
    *count = v.n;
    return DRS_OK;
  }
  return DRS_ERR_ARG;
}

int drs_params_get(drs_net_t* n, const char* name, const char* slot, float* host_dst, size_t count, void* stream) {
  float* src; size_t cnt;
  DRS_TRY(var_span(n, name, slot, &src, &cnt));
  if (!host_dst || count != cnt) return DRS_ERR_ARG;
  if (hipMemcpyAsync(host_dst, src, cnt * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return DRS_ERR_HIP;
  return drs_check(hipStreamSynchronize((hipStream_t)stream));
}

int drs_params_set(drs_net_t* n, const char* name, const char* slot, const float* host_src, size_t count, void* stream) {
  float* dst; size_t cnt;
  DRS_TRY(var_span(n, name, slot, &dst, &cnt));
  if (!host_src || count != cnt) return DRS_ERR_ARG;
  if (hipMemcpyAsync(dst, host_src, cnt * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return DRS_ERR_HIP;
  return drs_check(hipStreamSynchronize((hipStream_t)stream));
}

long long drs_net_global_step(drs_net_t* n, long long set_to) {
  if (!n) return -1;
  if (set_to >= 0) n->global_step = set_to;
  return n->global_step;
}

int drs_net_set_comm(drs_net_t* n, int world, int rank, drs_allreduce_fn allreduce, drs_wait_fn wait, void* user) {
  if (!n || world < 1 || rank < 0 || rank >= world || (world > 1 && !allreduce)) return DRS_ERR_ARG;
  release_rccl(n);         // the callback replaces library-side collectives (their streams and events go; the communicators are the caller's)
  n->world = world; n->rank = rank; n->allreduce = allreduce; n->wait = wait; n->comm_user = user;
  return DRS_OK;
}

// Data parallelism with the collectives issued by the library itself (RCCL through dlopen, rccl_comm.hip): comm_small / comm_big
// are ncclComm_t of `world` ranks on this process's GPU (drs_rccl_comm_create, or the host's own from the same librccl);
// comm_big may be NULL (everything then goes through comm_small); comm_stream (hipStream_t, may be NULL: the library then
// creates one) carries the gradient buckets.  The communicators stay the caller's: destroy them after the net.
// Passing comm_small = NULL removes library-side collectives: a net that had them returns to the single-rank form; a net on the
// callback (drs_net_set_comm) or without any communicator is left as it is.
// Default form (r04): INLINE -- comm_big is ignored; every sum is issued on the compute stream itself, in program order: the 16 sync-BN
// sums where they are needed, the whole gradient buffer as ONE all-reduce after the last filter gradient, then the loss and the
// confusion matrix.  No side stream, no event, one communicator driven from one stream.  Measured at world 1 with every collective
// forced on (B = 16, S = 64; profiles/r04/collectives_forms_world1.txt): no collectives 6.97 ms, inline 6.96 (the two-stream backward
// pass stays on: its filter-gradient stream never touches the communicator), the asynchronous form below 7.35 -- and 7.36 with the RCCL
// calls themselves skipped: what the asynchronous form costs is its own cross-stream event
// hand-overs (~25 of them a step), which is about what overlapping 8 MB of gradient buckets and eight 2 KB sums could save on 8 GPUs.
// DRS_RCCL_ASYNC=1 in the environment (read here) selects that form: backward sync-BN sums on a side stream under the filter
// gradient of the block above, gradient buckets on comm_big's stream as the layers finish.
// The form of the library-side collectives, read from the environment in ONE place (the host asks before it makes communicators, so
// that it makes as many as the library will use and labels its run with what really runs): DRS_RCCL_ASYNC != 0 -> asynchronous
// (two communicators); else DRS_RCCL_BUCKETS >= 2 -> inline + two overlapped gradient buckets (two communicators); else inline (one).
int drs_rccl_form(void) {
  const char* a = std::getenv("DRS_RCCL_ASYNC");
  if (a && std::atoi(a) != 0) return DRS_RCCL_FORM_ASYNC;
  const char* b = std::getenv("DRS_RCCL_BUCKETS");
  if (b && std::atoi(b) >= 2) return DRS_RCCL_FORM_BUCKETS;
  return DRS_RCCL_FORM_INLINE;
}

static int set_rccl_impl(drs_net_t* n, int world, int rank, void* comm_small, void* comm_big, void* comm_stream) {
  if (!n || world < 1 || rank < 0 || rank >= world) return DRS_ERR_ARG;
  const bool had = n->rccl_small != nullptr;
  release_rccl(n);
  if (!comm_small) { if (had) { n->world = 1; n->rank = 0; } return DRS_OK; }
  if (!drs_rccl_available()) return DRS_ERR_ARG;
  const int form = drs_rccl_form();
  n->rccl_inline = form != DRS_RCCL_FORM_ASYNC;
  n->rccl_buckets = form == DRS_RCCL_FORM_BUCKETS && comm_big != nullptr;
  if (n->rccl_inline && !n->rccl_buckets) comm_big = nullptr;
  // asynchronous sums in flight between two waits: one per block (backward sync-BN) + one per two blocks (gradient buckets) + 4
  n->comm_ring = std::max(64, 2 * (int)n->layers.size() + 8);
  n->comm_in_flight = 0;
  // the collectives' side streams at the LOWEST stream priority: a pool of hardware queues of their own, never the compute stream's
  // queue nor the filter-gradient stream's (train_step_impl: wg_stream at the highest; there for why)
  int comm_prio = 0, greatest_prio = 0;
  if (hipDeviceGetStreamPriorityRange(&comm_prio, &greatest_prio) != hipSuccess) comm_prio = 0;
#ifdef DRS_DEV
  if (g_wg_stream_prio == 1) comm_prio = greatest_prio; else if (g_wg_stream_prio == 0) comm_prio = 0;
#endif
  if (n->rccl_buckets) {      // one side stream for the two gradient buckets, three events
    for (auto& e : n->ev_bucket)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return DRS_ERR_HIP;
    if (comm_stream) { n->comm_stream = (hipStream_t)comm_stream; n->own_comm_stream = false; }
    else {
      if (stream_at_priority(&n->comm_stream, comm_prio) != hipSuccess) return DRS_ERR_HIP;
      n->own_comm_stream = true;
    }
  }
  if (!n->rccl_inline) {      // (the inline form needs no stream and no event of its own)
    n->comm_events.resize(2 * (size_t)n->comm_ring);
    for (auto& e : n->comm_events)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return DRS_ERR_HIP;
    if (stream_at_priority(&n->small_stream, comm_prio) != hipSuccess) return DRS_ERR_HIP;
    if (comm_stream) { n->comm_stream = (hipStream_t)comm_stream; n->own_comm_stream = false; }
    else {
      if (stream_at_priority(&n->comm_stream, comm_prio) != hipSuccess) return DRS_ERR_HIP;
      n->own_comm_stream = true;
    }
  }
  n->world = world; n->rank = rank; n->rccl_small = comm_small; n->rccl_big = comm_big; n->comm_next = 0;
  n->allreduce = nullptr; n->wait = nullptr; n->comm_user = nullptr;
  return DRS_OK;
}

// The backward pass of a training step on two streams (the filter gradients beside the batch-norm-backward / input-gradient chain,
// train_step_impl): mode -1 = by the library's rule (small steps only; the default), 0 = never (a host that must keep every launch
// of the step on ITS stream), 1 = always.  Same kernels on the same operands in every mode: bitwise the same step.
int drs_net_set_two_streams(drs_net_t* n, int mode) {
  if (!n || mode < -1 || mode > 1) return DRS_ERR_ARG;
  n->two_stream_mode = mode;
  return DRS_OK;
}

int drs_net_timing(drs_net_t* n, int enable) {
  if (!n) return DRS_ERR_ARG;
  n->timing = enable != 0;
  return DRS_OK;
}

// sums of the launches recorded since the last call, per kernel family; synchronises the events it reads
int drs_net_timing_summary(drs_net_t* n, int kind, char* name, int name_cap, int* launches, double* ms, double* work) {
  if (!n || kind < 0 || kind >= K_NKIND) return DRS_ERR_ARG;
  if (name && name_cap > 0) { std::strncpy(name, kKindNames[kind], name_cap - 1); name[name_cap - 1] = 0; }
  int cnt = 0; double tms = 0.0, w = 0.0;
  for (auto& r : n->recs) {
    if (r.kind != kind) continue;
    float t = 0.f;
    (void)hipEventSynchronize(r.e1);
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    ++cnt; tms += t; w += r.work;
  }
  if (launches) *launches = cnt;
  if (ms) *ms = tms;
  if (work) *work = w;
  if (kind == K_NKIND - 1) {       // the last family closes a summary: drop the records, keep their events for the next ones
    for (auto& r : n->recs) { n->pool_events.push_back(r.e0); n->pool_events.push_back(r.e1); }
    n->recs.clear();
  }
  return DRS_OK;
}

int drs_net_num_timing_kinds(void) { return K_NKIND; }

// is_training=False pass over the slab filled by drs_crop_normalize: pred (and logits when DRS_WANT_LOGITS); with DRS_WITH_LABELS
// the confusion matrix of (labels, pred) is ADDED into conf (validation, isprs:1599), gated by acc_mask under DRS_USE_ACC_MASK
static int forward_impl(drs_net_t* n, int B, int S, int flags, int ignore_label, void* stream) {
  if (!n || !check_bs(n, B, S) || !resolve(n)) return DRS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long M = (long long)B * S * S;
  const Ptrs& P = n->ptrs;
  DRS_TRY(forward_layers(n, B, S, false, (double)M, st));
  const Slab& f = n->slabs[n->feat];
  float* params = P.params;
  DRS_TRY(drs_classifier_loss(P.feat_act, B, S, f.P, f.C, 0, n->c_last, n->K, params + n->cls_w, params + n->cls_b, nullptr,
                              nullptr, nullptr, 0.f, (flags & DRS_WANT_LOGITS) ? P.logits : nullptr, P.pred,
                              nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, st));
  if (flags & DRS_WITH_LABELS)
    DRS_TRY(drs_confusion(P.labels, P.pred, (flags & DRS_USE_ACC_MASK) ? P.acc_mask : nullptr, (size_t)M, n->K, ignore_label, P.conf, st));
  return DRS_OK;
}

float drs_net_learning_rate(const drs_net_t* n, float lr0) {
  // tf.train.exponential_decay(lr0, global_step, 50000, factor, staircase=True) (isprs:1686)
  return n ? (float)(lr0 * std::pow((double)n->lr_decay, (double)(n->global_step / LR_DECAY_STEPS))) : 0.f;
}

int drs_apply_update(drs_net_t* n, float lr0, void* stream) {
  if (!n || !resolve(n)) return DRS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  {
    Timed t(n, st, K_MOMENTUM, n->n_params * 20.0);
    DRS_TRY(drs_momentum_update(n->ptrs.params, n->ptrs.grads, n->ptrs.momentum, n->n_params, n->n_decay,
                                drs_net_learning_rate(n, lr0), n->wd, MOMENTUM, 1.0f, st));
  }
  n->global_step += 1;
  return DRS_OK;
}

// One optimisation step on the slab / labels / masks currently in the bound buffers (net.py train_step).  Leaves, on the device:
// scalars[0] = mean CE over the pixels of ALL ranks, scalars[1] = 0.5 * sum w^2 (total loss = scalars[0] + wd * scalars[1],
// isprs:1089-1099, 646-651), pred, conf (this step, all ranks), and the updated variables unless DRS_NO_UPDATE.
// global_pixels = number of pixels the loss averages over on all ranks (<= 0: B*S*S*world; the contest form passes the number of
// unmasked pixels).  Every rank must hold the same B (the batch-norm count is B*S*S*world).
static int train_step_impl(drs_net_t* n, int B, int S, float lr0, int flags, double global_pixels, void* stream) {
  if (!n || !check_bs(n, B, S) || !resolve(n)) return DRS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long M = (long long)B * S * S;
  const Ptrs& P = n->ptrs;
  const double n_bn = (double)M * n->world;
  const double n_glob = global_pixels > 0 ? global_pixels : n_bn;
  n->comm_in_flight = 0;      // (a step that failed midway must not leave the next one short of event slots)
  // Small steps (the per-rank batches of data parallelism) run every convolution launch as one round: the chip drains and refills
  // between two dependent kernels.  The filter gradient of block i+1 depends only on that block's gz, so it goes to a stream of its
  // own and runs beside the batch-norm backward / input gradient of block i (two gz slabs in turn): its workgroups fill the CUs the
  // other chain's tails leave idle.  Same kernels, same operands: results are bitwise those of the one-stream order.  Not with the
  // host-callback collectives (the host's communicator is ordered against ITS stream, not this one), not while launches are timed
  // (per-kernel figures want kernels alone on the chip), and not from 2^18 pixels (measured: nothing to gain at B >= 64).
  bool two = n->two_stream_mode > 0 || (n->two_stream_mode < 0 && M < (1 << 18));
  // (inline collectives all run on `st`: the filter-gradient stream never touches the communicator, so the two go together;
  //  the asynchronous form with a single communicator would drive it from both streams)
  if (n->allreduce || n->timing || (n->rccl_small && !n->rccl_inline && !n->rccl_big)) two = false;
  const bool inline_comm = n->rccl_small && n->rccl_inline;
  if (two && !n->wg_stream) {
    // The filter gradients' stream at ANOTHER stream priority than the caller's -- not for the priority: the runtime keeps a pool of
    // hardware queues per priority level and deals the streams of a level out to its pool round robin, so a stream of the caller's
    // level can land on the very queue the compute stream uses.  The two "streams" are then one queue, nothing overlaps and the
    // cross-stream waits of this pass only add barriers: one stream more or less made before this one decides it (seen with the
    // collectives' library exchanged, profiles/r05/collectives_wire_time_model.txt: one rank's step 6.83 instead of 6.67 ms at
    // 16 x 64 x 64, 3.97 instead of 3.71 at S = 45 -- worse than ONE stream).  A stream of another level is never on the compute
    // stream's queue.  The highest level, the collectives' side streams (set_rccl_impl) the lowest: in-process A/B, no collectives:
    // highest = a lucky stream of the caller's level, lowest +1.5 % at S = 64; with every all-reduce a real launch both the same.
    // (development library, drs_debug_wg_stream_prio: 0 = the caller's level as before round 5, 1 = lowest and the collectives' streams highest)
    int least = 0, greatest = 0;
#ifdef DRS_DEV
    const int arm = g_wg_stream_prio;
#else
    const int arm = 2;
#endif
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    bool ok = stream_at_priority(&n->wg_stream, arm == 0 ? 0 : (arm == 1 ? least : greatest)) == hipSuccess;
    for (int i = 0; ok && i < 2; ++i)
      ok = hipEventCreateWithFlags(&n->ev_gz[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&n->ev_wg[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&n->ev_cls, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&n->ev_prep, hipEventDisableTiming) == hipSuccess;
    if (!ok) return DRS_ERR_HIP;
  }
  hipStream_t ws = two ? n->wg_stream : st;       // the stream the filter gradients run on
#ifdef DRS_DEV
  // r06 experiment (see g_wg_streams): per-layer gz / split slabs, 1..4 filter-gradient streams, eager or deferred issue
  const bool xs = two && n->layer_slabs && !collectives(n);
  const int NS = xs ? std::min(4, std::max(1, g_wg_streams)) : 1;
  const int defer = xs ? g_wg_defer : 0;
  if (xs) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    n->wg_streams[0] = n->wg_stream;
    for (int j = 1; j < NS; ++j)
      if (!n->wg_streams[j] && stream_at_priority(&n->wg_streams[j], greatest) != hipSuccess) return DRS_ERR_HIP;
    for (int j = 0; j < 4; ++j)
      if (!n->ev_wgS[j] && hipEventCreateWithFlags(&n->ev_wgS[j], hipEventDisableTiming) != hipSuccess) return DRS_ERR_HIP;
    for (int j = 0; j < 16; ++j)
      if (!n->ev_gzL[j] && hipEventCreateWithFlags(&n->ev_gzL[j], hipEventDisableTiming) != hipSuccess) return DRS_ERR_HIP;
    if (!n->ev_chain_end && hipEventCreateWithFlags(&n->ev_chain_end, hipEventDisableTiming) != hipSuccess) return DRS_ERR_HIP;
  }
#else
  constexpr bool xs = false;
  constexpr int defer = 0;
#endif
  float* params = P.params;
  float* grads = P.grads;
  const int nL = (int)n->layers.size();
  unsigned int* conf = P.conf;
  // one launch: the flipped / transposed filters of the input-gradient passes, the confusion matrix zeroed, and the gradients of
  // the conv biases zeroed (they sit in front of a mean-subtracting batch norm: their gradient is identically zero).  Nothing of
  // the forward pass needs it: with two streams it runs on the filter-gradient stream BESIDE the forward pass (behind everything
  // the compute stream held when the step began), and the classifier waits for it
  auto step_prep = [&](hipStream_t sp) -> int {
    StepPrepArgs pa;
    pa.n = 0;
    for (int i = 1; i < nL; ++i) {
      const Layer& L = n->layers[i];
      if (pa.n == STEP_PREP_MAX) {                       // (deeper nets than any of the reference's: the rest one by one)
        DRS_TRY(drs_filter_flip_transpose(params + L.w_off, P.L[i].wt, L.k, L.cin, L.cout, sp));
        continue;
      }
      pa.w[pa.n] = params + L.w_off; pa.wt[pa.n] = P.L[i].wt;
      pa.k[pa.n] = L.k; pa.cin[pa.n] = L.cin; pa.cout[pa.n] = L.cout;
      ++pa.n;
    }
    pa.z0 = conf; pa.nz0 = n->K * n->K;
    pa.z1 = grads + n->first_bias; pa.nz1 = (int)(n->cls_b - n->first_bias);
    DRS_TRY(drs_step_prep(pa, sp));
    return DRS_OK;
  };
#ifdef DRS_DEV
  const bool side_jobs = two && !g_reductions_on_chain;      // (A/B, drs_debug_reductions_on_chain(1): they stay on `st`, as before round 5)
#else
  const bool side_jobs = two;
#endif
  if (side_jobs) {
    if (hipEventRecord(n->ev_cls, st) != hipSuccess || hipStreamWaitEvent(ws, n->ev_cls, 0) != hipSuccess) return DRS_ERR_HIP;
    DRS_JITTER(ws);
    DRS_TRY(step_prep(ws));
    if (hipEventRecord(n->ev_prep, ws) != hipSuccess) return DRS_ERR_HIP;
  }
  DRS_TRY(forward_layers(n, B, S, true, n_bn, st));
  if (side_jobs) { if (hipStreamWaitEvent(st, n->ev_prep, 0) != hipSuccess) return DRS_ERR_HIP; }
  else DRS_TRY(step_prep(st));
  // classifier + loss + gradient wrt the features
  const Slab& f = n->slabs[n->feat];
  float* gfeat = P.feat_gact;
  double* scalars = P.scalars;
  double* scratch = P.colsum_scratch;
  {
    Timed t(n, st, K_CLS, M * n->c_last * 8.0);
    DRS_TRY(drs_classifier_loss(P.feat_act, B, S, f.P, f.C, 0, n->c_last, n->K, params + n->cls_w, params + n->cls_b,
                                P.labels, (flags & DRS_USE_LOSS_MASK) ? P.loss_mask : nullptr,
                                (flags & DRS_USE_ACC_MASK) ? P.acc_mask : nullptr, (float)(1.0 / n_glob),
                                (flags & DRS_WANT_LOGITS) ? P.logits : nullptr, P.pred, gfeat, f.C, 0,
                                P.dw_partial, P.db_partial, P.loss_partial, conf, st));
  }
  // the classifier's slab reductions (its kernel / bias gradients, the cross-entropy sum) and the L2 term: seven launches of ~5 us that
  // nothing needs before the end of the step -- in the two-stream backward pass they go to the filter-gradient stream (below), off
  // the chain the step waits for
  const int crow = drs_classifier_rows(B, S);
  auto slab_reductions = [&](hipStream_t s) -> int {
    DRS_TRY(drs_rows_reduce_f32(P.dw_partial, crow, n->c_last * n->K, grads + n->cls_w, scratch, s));
    DRS_TRY(drs_rows_reduce_f32(P.db_partial, crow, n->K, grads + n->cls_b, scratch, s));
    DRS_TRY(drs_sum_f64(P.loss_partial, crow, scalars, s));
    DRS_TRY(drs_l2_loss(params, n->n_decay, P.l2_scratch, scalars + 1, s));
    return DRS_OK;
  };
  // gradient all-reduce in buckets that overlap the rest of the backward pass: kernel gradients go as their layers finish, last
  // layers first; the small classifier / SE / bias tail goes last
  std::vector<int> pending;      // (handles of asynchronous sums: filled only by the callback / asynchronous forms; empty, no allocation, otherwise)
  size_t bucket_hi = n->cls_w;
  char written[MAX_SLABS] = {0};
  written[n->feat] = 1;
  float* gzb[2] = {P.gz[0], P.gz[1]};
  float* gxh = P.gxh;
  float* partial = P.partial;
  double* sums = P.sums;
  float* bwd_means = P.bwd_means;

  if (side_jobs) {
    if (hipEventRecord(n->ev_cls, st) != hipSuccess || hipStreamWaitEvent(ws, n->ev_cls, 0) != hipSuccess) return DRS_ERR_HIP;
    DRS_JITTER(ws);
    DRS_TRY(slab_reductions(ws));           // (the step joins `ws` before it reads the gradient buffer or the loss: the filter gradients' events)
  } else {
    DRS_TRY(slab_reductions(st));
  }
  // two streams: what this thread enqueues on `st` from here to the end of the backward pass is the chain the step waits for, beside
  // the filter gradients on `ws` (drs_common.hpp: its waves take the top priority; the filter gradient's launches clear the hint)
  struct ChainHint { int old; explicit ChainHint(int v) : old(drs_tl_chain) { drs_tl_chain = v; } ~ChainHint() { drs_tl_chain = old; } };
  ChainHint chain_hint(two ? (collectives(n) ? 2 : 1) : 0);
  // DRS_RCCL_BUCKETS=2 (opt-in until an 8-GPU A/B exists): the gradient buffer as two all-reduces on a side stream with a communicator
  // of its own; everything else as in the inline form
  const bool buckets = inline_comm && n->rccl_buckets && n->rccl_big;
  bool bucket_sent = false;
  size_t bucket_lo = 0;
  int bucket_layer = 0;
  if (buckets) {
    size_t total = 0, acc = 0;
    for (int i = 0; i < nL; ++i) total += (size_t)n->layers[i].k * n->layers[i].k * n->layers[i].cin * n->layers[i].cout;
    for (int i = nL - 1; i >= 0; --i) {
      acc += (size_t)n->layers[i].k * n->layers[i].k * n->layers[i].cin * n->layers[i].cout;
      if (10 * acc >= 7 * total) { bucket_layer = i; break; }
    }
  }
  auto bucket_sum = [&](float* ptr, size_t cnt) -> int {
    Timed t(n, n->comm_stream, K_AR_GRAD, 4.0 * (double)cnt);
    return drs_rccl_all_reduce_sum(n->rccl_big, ptr, cnt, F32, n->comm_stream);
  };

  auto filter_gradient = [&](int i) -> int {
    const Layer& L = n->layers[i];
    const Slab& in = n->slabs[L.src];
    ChainHint not_chain(0);
#ifdef DRS_DEV
    if (xs) {      // this layer's own slabs, stream i % NS, behind this layer's gz (or the chain's end)
      hipStream_t wsi = n->wg_streams[i % NS];
      if (hipStreamWaitEvent(wsi, defer == 1 ? n->ev_chain_end : n->ev_gzL[i], 0) != hipSuccess) return DRS_ERR_HIP;
      DRS_TRY(drs_conv_wgrad(P.L[i].act_in, B, S, in.P, in.C, 0, P.L[i].gzL, L.halo, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k, L.cin, L.cout,
                             P.L[i].slabL, grads + L.w_off, wsi));
      if (hipEventRecord(n->ev_wgS[i % NS], wsi) != hipSuccess) return DRS_ERR_HIP;
      return DRS_OK;
    }
#endif
    if (two && hipStreamWaitEvent(ws, n->ev_gz[i & 1], 0) != hipSuccess) return DRS_ERR_HIP;      // block i's gz is written
    if (two) DRS_JITTER(ws);
    {
      Timed t(n, ws, K_CONV_WGRAD, 2.0 * M * L.k * L.k * L.cin * L.cout);
      DRS_TRY(drs_conv_wgrad(P.L[i].act_in, B, S, in.P, in.C, 0, gzb[two ? (i & 1) : 0], L.halo, L.cout, 0, L.k, L.rate, L.pad_b, L.cin_k,
                             L.cin, L.cout, P.slab, grads + L.w_off, ws));
    }
    if (collectives(n) && !inline_comm && i > 0 && (nL - i) % 2 == 0) {      // every second layer: one bucket
      int h;
      DRS_TRY(all_reduce(n, K_AR_GRAD, grads + L.w_off, bucket_hi - L.w_off, F32, 1, ws, &h, two));
      pending.push_back(h);
      bucket_hi = L.w_off;
    }
    if (two) DRS_JITTER(ws);
    if (two && hipEventRecord(n->ev_wg[i & 1], ws) != hipSuccess) return DRS_ERR_HIP;             // this gz slab may be rewritten
    if (buckets && i == bucket_layer && i > 0) {
      // first bucket: the kernels of blocks bucket_layer .. last (Dilated8Pooling: conv5 .. conv8, 76 % of the gradient bytes), summed on the
      // big communicator's stream while blocks bucket_layer-1 .. 0 are still in their backward pass
      if (hipEventRecord(n->ev_bucket[0], ws) != hipSuccess || hipStreamWaitEvent(n->comm_stream, n->ev_bucket[0], 0) != hipSuccess) return DRS_ERR_HIP;
      DRS_TRY(bucket_sum(grads + L.w_off, n->cls_w - L.w_off));
      bucket_sent = true;
      bucket_lo = L.w_off;
    }
    return DRS_OK;
  };

  int deferred = -1;      // block whose filter gradient is still to be computed (it only needs that block's gz, still in place)
  for (int i = nL - 1; i >= 0; --i) {
    const Layer& L = n->layers[i];
    const LayerPtrs& q = P.L[i];
    const Slab& out = n->slabs[L.dst];
    const float* gcur = q.gact_out;
    const bool mx = L.pool == 1;
    const float* gsrc = gcur;
    int ldg = out.C, cg = L.dst_coff;
    if (L.se >= 0) {
      const SeBlock& s = n->se[L.se];
      Timed t(n, st, K_SE_BWD, M * L.cout * 16.0);
      DRS_TRY(drs_se_backward(gcur, out.C, L.dst_coff, q.se_act, q.se_s, q.se_e1, q.se_e2, params + s.w1, params + s.w2, B, S, L.cout, s.R, P.gpool,
                              grads + s.w1, grads + s.b1, grads + s.w2, grads + s.b2, P.se_scratch, st));
      gsrc = P.gpool; ldg = L.cout; cg = 0;
    } else if (L.pool == 2) {
      Timed t(n, st, K_AVG_BWD, M * L.cout * 8.0);
      DRS_TRY(drs_avg_pool_backward(gcur, out.C, L.dst_coff, B, S, L.cout, L.avg_k, P.gpool, st));
      gsrc = P.gpool; ldg = L.cout; cg = 0;
    }
    float* z = q.z;
    float* mr = q.mean_rstd;
    {
      Timed t(n, st, K_BN_BWD_REDUCE, M * L.cout * (mx ? 13.0 : 12.0));
      DRS_TRY(drs_bn_backward_reduce(gsrc, ldg, cg, z, mx ? q.idx : nullptr, B, S, L.cout, mr, n->alpha, mx ? 1 : 0, gxh,
                                     partial, st));
    }
    // single rank: the sums need no all-reduce, so the reduction also leaves the two means the apply pass subtracts (as fp32, its own
    // expressions: the same bits) and that pass does no fp64 division per workgroup
    const bool means_form = !collectives(n);
    if (means_form) DRS_TRY(drs_stats_reduce_means(partial, drs_bn_backward_rows(B, S, L.cout, mx ? 1 : 0), L.cout, n_bn, sums, bwd_means, st));
    else DRS_TRY(drs_stats_reduce(partial, drs_bn_backward_rows(B, S, L.cout, mx ? 1 : 0), L.cout, sums, nullptr, st));
    // sync batch norm: the all-reduce of (sum g, sum g*xhat) runs on the collective's stream while this stream computes the
    // filter gradient of the block above
    // (two streams: the filter gradient runs beside this chain anyway, so the sum goes on this stream itself -- no event hand-over
    // on the critical path, and the small communicator is then driven from this stream only, forward and backward)
    int h_bn;
    DRS_TRY(all_reduce(n, K_AR_SYNCBN, sums, 2 * (size_t)L.cout, F64, two ? 0 : 1, st, &h_bn));
    if (deferred >= 0 && (defer == 0 || defer == 3)) DRS_TRY(filter_gradient(deferred));
    DRS_TRY(wait_handles(n, &h_bn, 1, st));
    float* gz = gzb[two ? (i & 1) : 0];
#ifdef DRS_DEV
    if (xs) gz = q.gzL;
#endif
    if (two && !xs && i + 2 < nL && hipStreamWaitEvent(st, n->ev_wg[i & 1], 0) != hipSuccess) return DRS_ERR_HIP;   // block i+2's filter gradient has read this slab
    if (two) DRS_JITTER(st);
    {
      Timed t(n, st, K_BN_BWD_APPLY, M * L.cout * 12.0);
      if (means_form) DRS_TRY(drs_bn_backward_apply_means(gxh, z, B, S, L.cout, mr, bwd_means, gz, L.halo, L.cout, 0, st));
      else DRS_TRY(drs_bn_backward_apply(gxh, z, B, S, L.cout, mr, sums, n_bn, gz, L.halo, L.cout, 0, st));
    }
#ifdef DRS_DEV
    if (xs && defer != 3 && hipEventRecord(n->ev_gzL[i], st) != hipSuccess) return DRS_ERR_HIP;
#endif
    if (two && !xs && hipEventRecord(n->ev_gz[i & 1], st) != hipSuccess) return DRS_ERR_HIP;
    if (two) DRS_JITTER(st);
    if (L.src != 0) {
      const Slab& in = n->slabs[L.src];
      const int acc = written[L.src] ? 1 : 0;
      written[L.src] = 1;
      Timed t(n, st, K_CONV_DGRAD, 2.0 * M * L.k * L.k * L.cin * L.cout);
      DRS_TRY(drs_conv_forward_ws(gz, B, S, L.halo, L.cout, 0, q.wt, nullptr, L.k, L.rate, L.pad_a, L.cout, L.cin,
                                  q.gact_in, in.C, 0, acc, nullptr, P.conv_ws, P.conv_ws_floats, st));
    }
#ifdef DRS_DEV
    // defer 3: block i's filter gradient is released by the END of block i's input gradient (not by its gz): it then runs beside block
    // i-1's elementwise passes and input gradient instead of beside its own block's input gradient
    if (xs && defer == 3 && hipEventRecord(n->ev_gzL[i], st) != hipSuccess) return DRS_ERR_HIP;
#endif
    deferred = i;
  }
#ifdef DRS_DEV
  if (xs) {
    if (defer == 1 || defer == 2) {
      if (hipEventRecord(n->ev_chain_end, st) != hipSuccess) return DRS_ERR_HIP;
      for (int i = nL - 1; i >= 0; --i) DRS_TRY(filter_gradient(i));
    } else {
      DRS_TRY(filter_gradient(deferred));
    }
    // the side jobs went to wg_stream = wg_streams[0]: its last event covers them when a filter gradient followed; join every stream used
    for (int j = 0; j < std::min(NS, nL); ++j)
      if (hipStreamWaitEvent(st, n->ev_wgS[j], 0) != hipSuccess) return DRS_ERR_HIP;
  } else
#endif
  {
  DRS_TRY(filter_gradient(deferred));
  if (two) {      // every filter gradient is in place before the rest of the step reads the gradient buffer
    if (hipStreamWaitEvent(st, n->ev_wg[0], 0) != hipSuccess) return DRS_ERR_HIP;
    if (nL > 1 && hipStreamWaitEvent(st, n->ev_wg[1], 0) != hipSuccess) return DRS_ERR_HIP;
  }
  }
  if (inline_comm) {      // one all-reduce of the whole flat gradient buffer, then the loss and the confusion matrix, in program order
    if (buckets) {
      // second bucket: the earlier layers' kernels and the classifier / SE / bias tail, behind the first on the big communicator's
      // stream; then ONE wait.  Three event hand-overs a step in all (the asynchronous form of r03 had ~25).
      if (hipEventRecord(n->ev_bucket[1], st) != hipSuccess || hipStreamWaitEvent(n->comm_stream, n->ev_bucket[1], 0) != hipSuccess) return DRS_ERR_HIP;
      if (!bucket_sent) DRS_TRY(bucket_sum(grads, n->cls_w));
      else if (bucket_lo > 0) DRS_TRY(bucket_sum(grads, bucket_lo));
      DRS_TRY(bucket_sum(grads + n->cls_w, n->n_params - n->cls_w));
      if (hipEventRecord(n->ev_bucket[2], n->comm_stream) != hipSuccess || hipStreamWaitEvent(st, n->ev_bucket[2], 0) != hipSuccess) return DRS_ERR_HIP;
    } else {
      DRS_TRY(all_reduce(n, K_AR_GRAD, grads, n->n_params, F32, 0, st, nullptr));
    }
    DRS_TRY(all_reduce(n, K_AR_SCALARS, scalars, 1, F64, 0, st, nullptr));
    DRS_TRY(all_reduce(n, K_AR_SCALARS, conf, (size_t)n->K * n->K, I32, 0, st, nullptr));
  } else if (collectives(n)) {
    int h;
    DRS_TRY(all_reduce(n, K_AR_GRAD, grads, bucket_hi, F32, 1, st, &h, two));                         // the remaining (earliest) layers
    pending.push_back(h);
    DRS_TRY(all_reduce(n, K_AR_GRAD, grads + n->cls_w, n->n_params - n->cls_w, F32, 1, st, &h, two)); // classifier, SE layers and every bias (small)
    pending.push_back(h);
    // (library-side RCCL: one communicator is never driven from two streams that no event orders.  One-stream backward pass: the
    // backward phase's small sums all go through the small communicator's side stream, in order.  Two-stream backward pass: the
    // small communicator is driven from the compute stream alone -- the backward sync-BN sums above, these two -- and every
    // asynchronous sum, the small tail bucket included, went to the big communicator's stream: `big_only`.)
    const int side = (n->rccl_small && !two) ? 1 : 0;
    DRS_TRY(all_reduce(n, K_AR_SCALARS, scalars, 1, F64, side, st, &h));
    pending.push_back(h);
    DRS_TRY(all_reduce(n, K_AR_SCALARS, conf, (size_t)n->K * n->K, I32, side, st, &h));
    pending.push_back(h);
    DRS_TRY(wait_handles(n, pending.data(), pending.size(), st));
  }
  DRS_TRY(drs_scale_f64(scalars, 1, 1.0 / n_glob, st));
  if (!(flags & DRS_NO_UPDATE)) DRS_TRY(drs_apply_update(n, lr0, st));
  return DRS_OK;
}

// the entry points whose host side allocates (strings, vectors): an allocation failure is a status, nothing throws across the ABI
int drs_forward(drs_net_t* n, int B, int S, int flags, int ignore_label, void* stream) {
  try { return forward_impl(n, B, S, flags, ignore_label, stream); } catch (...) { return DRS_ERR_HIP; }
}

int drs_train_step(drs_net_t* n, int B, int S, float lr0, int flags, double global_pixels, void* stream) {
  try { return train_step_impl(n, B, S, lr0, flags, global_pixels, stream); } catch (...) { return DRS_ERR_HIP; }
}

int drs_net_set_rccl(drs_net_t* n, int world, int rank, void* comm_small, void* comm_big, void* comm_stream) {
  try { return set_rccl_impl(n, world, rank, comm_small, comm_big, comm_stream); } catch (...) { return DRS_ERR_HIP; }
}

}  // extern "C"
