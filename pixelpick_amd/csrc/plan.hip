// plan.hip — native launch-list executor of the C ABI (include/pixelpick_hip.h: pp_plan_*).
//
// With static shapes and stable addresses every C-ABI call of a train step (model.py:101-122 of the reference: forward, loss,
// backward, optimiser) repeats with identical arguments.  The step is recorded ONCE - the entry points it calls with their
// converted arguments, the stream fork / join operations between the main and the weight-gradient queue, and "host breaks"
// where the caller has to do something itself (the RCCL all-reduces of a data-parallel step) - and re-issued from this file's
// loop: one foreign call per step instead of ~500, no interpreter between two launches.  Each recorded call goes through its
// entry point again (same planning, same kernels, same queues as an eager step: bit-identical results), so nothing here knows
// about kernels; what is removed is the host time around them.
//
// An argument travels as one 8-byte slot (integers sign-extended, pointers as addresses, floats in the low four bytes); a
// typed thunk per entry point - generated from the function's own signature by the template below, so a changed prototype
// cannot go out of step - unpacks the slots and makes an ordinary C++ call.
#include "pp_common.h"

#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

namespace pp {
namespace {

template <typename T>
inline T unpack_slot(uint64_t s)
{
    if constexpr (std::is_same_v<T, float>) {
        float f;
        uint32_t lo = (uint32_t)s;
        memcpy(&f, &lo, 4);
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double d;
        memcpy(&d, &s, 8);
        return d;
    } else if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(s));
    } else {
        static_assert(std::is_integral_v<T>, "C-ABI arguments are integers, floats or pointers");
        return static_cast<T>(static_cast<int64_t>(s));
    }
}

template <typename F, F Fn>
struct Thunk;
template <typename... A, int (*Fn)(A...)>
struct Thunk<int (*)(A...), Fn> {
    static constexpr int kArgs = (int)sizeof...(A);
    template <size_t... I>
    static int call_impl(const uint64_t* s, std::index_sequence<I...>) { return Fn(unpack_slot<A>(s[I])...); }
    static int call(const uint64_t* s) { return call_impl(s, std::index_sequence_for<A...>{}); }
};

struct Entry {
    const void* fn;
    int (*call)(const uint64_t*);
    int nargs;
    const char* name;
};

#define PP_PLAN_ENTRY(f) Entry{reinterpret_cast<const void*>(&f), &Thunk<decltype(&f), &f>::call, Thunk<decltype(&f), &f>::kArgs, #f}

// every entry point that enqueues work (workspace queries and debugging knobs are not part of a step)
const Entry kEntries[] = {
    PP_PLAN_ENTRY(pp_acq_score_topk),
    PP_PLAN_ENTRY(pp_acq_score_map),
    PP_PLAN_ENTRY(pp_acq_softmax_sum),
    PP_PLAN_ENTRY(pp_uncertainty_from_prob),
    PP_PLAN_ENTRY(pp_topk_select),
    PP_PLAN_ENTRY(pp_acq_lowres_score_topk),
    PP_PLAN_ENTRY(pp_acq_lowres_score_at),
    PP_PLAN_ENTRY(pp_conv2d_fwd),
    PP_PLAN_ENTRY(pp_conv2d_fwd_stats),
    PP_PLAN_ENTRY(pp_bn_train_fwd_partials),
    PP_PLAN_ENTRY(pp_conv2d_fwd_bn_act),
    PP_PLAN_ENTRY(pp_dwconv3x3_fwd_bn_act),
    PP_PLAN_ENTRY(pp_bn_finalize_partials),
    PP_PLAN_ENTRY(pp_dwconv3x3_fwd_fused),
    PP_PLAN_ENTRY(pp_dwconv3x3_bwd_weight_affine_in),
    PP_PLAN_ENTRY(pp_conv2d_fwd_affine_in),
    PP_PLAN_ENTRY(pp_conv2d_bwd_data),
    PP_PLAN_ENTRY(pp_conv2d_bwd_weight),
    PP_PLAN_ENTRY(pp_conv2d_fwd_bn_train),
    PP_PLAN_ENTRY(pp_conv2d_bwd_data_bn_bwd),
    PP_PLAN_ENTRY(pp_x3_split),
    PP_PLAN_ENTRY(pp_conv2d_fwd_pre),
    PP_PLAN_ENTRY(pp_conv2d_bwd_data_pre),
    PP_PLAN_ENTRY(pp_conv2d_bwd_data_multi),
    PP_PLAN_ENTRY(pp_conv2d_fwd_pre2),
    PP_PLAN_ENTRY(pp_conv2d_bwd_data_pre2),
    PP_PLAN_ENTRY(pp_x3_split_weights),
    PP_PLAN_ENTRY(pp_conv2d_bwd_weight_pre),
    PP_PLAN_ENTRY(pp_conv2d_bwd_weight_partials),
    PP_PLAN_ENTRY(pp_wgrad_reduce_batch),
    PP_PLAN_ENTRY(pp_bn_train_fwd),
    PP_PLAN_ENTRY(pp_bn_eval_affine),
    PP_PLAN_ENTRY(pp_scale_shift_act),
    PP_PLAN_ENTRY(pp_bn_bwd),
    PP_PLAN_ENTRY(pp_bn_train_fwd_fused),
    PP_PLAN_ENTRY(pp_dwconv3x3_bn_train_fwd_fused),
    PP_PLAN_ENTRY(pp_bn_bwd_fused),
    PP_PLAN_ENTRY(pp_bn_bwd_fused_sparse),
    PP_PLAN_ENTRY(pp_row_flags),
    PP_PLAN_ENTRY(pp_conv1x1_bwd_weight_sparse),
    PP_PLAN_ENTRY(pp_conv1x1_bwd_data_sparse),
    PP_PLAN_ENTRY(pp_dwconv3x3_fwd),
    PP_PLAN_ENTRY(pp_dwconv3x3_bwd_data),
    PP_PLAN_ENTRY(pp_dwconv3x3_bwd_weight),
    PP_PLAN_ENTRY(pp_dwconv3x3_bwd_weight_partials),
    PP_PLAN_ENTRY(pp_groupnorm_relu_fwd),
    PP_PLAN_ENTRY(pp_groupnorm_relu_bwd),
    PP_PLAN_ENTRY(pp_maxpool2d_fwd),
    PP_PLAN_ENTRY(pp_maxpool2d_bwd),
    PP_PLAN_ENTRY(pp_pad2d),
    PP_PLAN_ENTRY(pp_crop2d_add),
    PP_PLAN_ENTRY(pp_bilinear_fwd),
    PP_PLAN_ENTRY(pp_bilinear_bwd),
    PP_PLAN_ENTRY(pp_image_colsum),
    PP_PLAN_ENTRY(pp_image_broadcast),
    PP_PLAN_ENTRY(pp_dropout),
    PP_PLAN_ENTRY(pp_dropout2d),
    PP_PLAN_ENTRY(pp_aug_resample_h),
    PP_PLAN_ENTRY(pp_aug_vcrop),
    PP_PLAN_ENTRY(pp_aug_labels),
    PP_PLAN_ENTRY(pp_aug_jitter),
    PP_PLAN_ENTRY(pp_aug_blur),
    PP_PLAN_ENTRY(pp_aug_blur_q8),
    PP_PLAN_ENTRY(pp_aug_to_tensor),
    PP_PLAN_ENTRY(pp_sparse_ce_fwd_bwd),
    PP_PLAN_ENTRY(pp_sparse_ce_lowres_fwd_bwd),
    PP_PLAN_ENTRY(pp_confusion_matrix_update),
    PP_PLAN_ENTRY(pp_adam_step_flat),
    PP_PLAN_ENTRY(pp_sgd_step_flat),
    PP_PLAN_ENTRY(pp_add2d),
    PP_PLAN_ENTRY(pp_nhwc_to_nchw),
    PP_PLAN_ENTRY(pp_nchw_to_nhwc),
};

const Entry* find_entry(const void* fn)
{
    for (const Entry& e : kEntries)
        if (e.fn == fn) return &e;
    return nullptr;
}

enum OpKind : int { kCall = 0, kEventRecord = 1, kStreamWait = 2, kHostBreak = 3 };

struct Op {
    int kind;
    int nargs;
    const Entry* entry;     // kCall
    size_t slot0;           // kCall: first slot in Plan::slots
    void* a;                // kEventRecord: event;  kStreamWait: stream
    void* b;                // kEventRecord: stream; kStreamWait: event
};

}  // namespace

struct Plan {
    std::vector<Op> ops;
    std::vector<uint64_t> slots;
    std::vector<hipEvent_t> owned_events;
};

}  // namespace pp

using pp::Plan;

extern "C" {

pp_plan_t pp_plan_create(void) { return reinterpret_cast<pp_plan_t>(new (std::nothrow) Plan()); }

void pp_plan_destroy(pp_plan_t plan)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p) return;
    for (hipEvent_t e : p->owned_events) (void)hipEventDestroy(e);
    delete p;
}

int64_t pp_plan_size(pp_plan_t plan) { return plan ? (int64_t) reinterpret_cast<Plan*>(plan)->ops.size() : 0; }

int pp_plan_entry_args(const void* fn)
{
    const pp::Entry* e = pp::find_entry(fn);
    return e ? e->nargs : -1;
}

int pp_plan_add_call(pp_plan_t plan, const void* fn, const uint64_t* slots, int n_slots)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p || !fn || (n_slots > 0 && !slots)) return pp::fail(PP_ERR_BAD_ARG, "plan_add_call: null argument");
    const pp::Entry* e = pp::find_entry(fn);
    if (!e) return pp::fail(PP_ERR_UNSUPPORTED, "plan_add_call: %p is not an entry point of this library that enqueues work", fn);
    if (n_slots != e->nargs) return pp::fail(PP_ERR_BAD_ARG, "plan_add_call: %s takes %d arguments, %d slots given", e->name, e->nargs, n_slots);
    pp::Op op{pp::kCall, n_slots, e, p->slots.size(), nullptr, nullptr};
    p->slots.insert(p->slots.end(), slots, slots + n_slots);
    p->ops.push_back(op);
    return PP_OK;
}

int pp_plan_add_event_record(pp_plan_t plan, void* event, pp_stream_t stream)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p || !event) return pp::fail(PP_ERR_BAD_ARG, "plan_add_event_record: null argument");
    p->ops.push_back(pp::Op{pp::kEventRecord, 0, nullptr, 0, event, stream});
    return PP_OK;
}

int pp_plan_add_stream_wait(pp_plan_t plan, pp_stream_t stream, void* event)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p || !event) return pp::fail(PP_ERR_BAD_ARG, "plan_add_stream_wait: null argument");
    p->ops.push_back(pp::Op{pp::kStreamWait, 0, nullptr, 0, stream, event});
    return PP_OK;
}

int pp_plan_add_join(pp_plan_t plan, pp_stream_t waiting, pp_stream_t waited_for)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p) return pp::fail(PP_ERR_BAD_ARG, "plan_add_join: null plan");
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return pp::fail(PP_ERR_LAUNCH, "plan_add_join: hipEventCreateWithFlags failed");
    p->owned_events.push_back(ev);
    p->ops.push_back(pp::Op{pp::kEventRecord, 0, nullptr, 0, ev, waited_for});
    p->ops.push_back(pp::Op{pp::kStreamWait, 0, nullptr, 0, waiting, ev});
    return PP_OK;
}

int pp_plan_add_host_break(pp_plan_t plan)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p) return pp::fail(PP_ERR_BAD_ARG, "plan_add_host_break: null plan");
    p->ops.push_back(pp::Op{pp::kHostBreak, 0, nullptr, 0, nullptr, nullptr});
    return PP_OK;
}

int pp_plan_replay(pp_plan_t plan, int64_t from, int64_t* next)
{
    Plan* p = reinterpret_cast<Plan*>(plan);
    if (!p || !next || from < 0) return pp::fail(PP_ERR_BAD_ARG, "plan_replay: bad argument");
    const int64_t n = (int64_t)p->ops.size();
    const uint64_t* slots = p->slots.data();
    int64_t i = from;
    for (; i < n; ++i) {
        const pp::Op& op = p->ops[(size_t)i];
        switch (op.kind) {
        case pp::kCall: {
            const int rc = op.entry->call(slots + op.slot0);
            if (rc != PP_OK) {
                *next = i;
                return rc;            // pp_last_error() holds the entry point's own message
            }
            break;
        }
        case pp::kEventRecord:
            if (hipEventRecord(reinterpret_cast<hipEvent_t>(op.a), reinterpret_cast<hipStream_t>(op.b)) != hipSuccess) {
                *next = i;
                return pp::fail(PP_ERR_LAUNCH, "plan_replay: hipEventRecord failed at op %lld", (long long)i);
            }
            break;
        case pp::kStreamWait:
            if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(op.a), reinterpret_cast<hipEvent_t>(op.b), 0) != hipSuccess) {
                *next = i;
                return pp::fail(PP_ERR_LAUNCH, "plan_replay: hipStreamWaitEvent failed at op %lld", (long long)i);
            }
            break;
        default:                      // host break: the caller does its part (a collective) and resumes behind it
            *next = i + 1;
            return PP_OK;
        }
    }
    *next = n;
    return PP_OK;
}

}  // extern "C"
