// plan.hip — recorded launch sequences ("plans") and ABI housekeeping for libvmv_hip.so.
// A UNet forward is >700 launches; recording the argument blocks once and replaying them from one C call
// removes the per-launch Python/ctypes cost.  Replay is a plain in-order launch loop on the
// caller's stream, hence capturable into a hipGraph by the caller.
#include "common.h"
#include <vector>
#include <cstring>
#include <new>

struct VmvPlan {
    struct Op { int op; std::vector<unsigned char> args; };
    std::vector<Op> ops;
};

namespace {
int run_op(const VmvPlan::Op& o, void* stream) {
    switch (o.op) {
        case VMV_OP_GEMM: return vmv_gemm(reinterpret_cast<const VmvGemmParams*>(o.args.data()), stream);
        case VMV_OP_GN_STATS: return vmv_groupnorm_stats(reinterpret_cast<const VmvGroupNormParams*>(o.args.data()), stream);
        case VMV_OP_GN_APPLY: return vmv_groupnorm_apply(reinterpret_cast<const VmvGroupNormParams*>(o.args.data()), stream);
        case VMV_OP_LAYERNORM: return vmv_layernorm(reinterpret_cast<const VmvLayerNormParams*>(o.args.data()), stream);
        case VMV_OP_ATTENTION: return vmv_attention(reinterpret_cast<const VmvAttnParams*>(o.args.data()), stream);
        case VMV_OP_SOFTMAX: return vmv_softmax_rows(reinterpret_cast<const VmvSoftmaxParams*>(o.args.data()), stream);
        case VMV_OP_COPY: return vmv_permute_copy(reinterpret_cast<const VmvCopyParams*>(o.args.data()), stream);
        case VMV_OP_FF: return vmv_ff_fused(reinterpret_cast<const VmvFfParams*>(o.args.data()), stream);
        case VMV_OP_GN_TABLE: return vmv_groupnorm_table(reinterpret_cast<const VmvGroupNormParams*>(o.args.data()), stream);
        case VMV_OP_COMM: return vmv_comm_run(reinterpret_cast<const VmvCommParams*>(o.args.data()), stream);
        case VMV_OP_GN_FUSED: {
            const VmvGroupNormParams* g = reinterpret_cast<const VmvGroupNormParams*>(o.args.data());
            return vmv_groupnorm_fused(g, g->chunk_rows, stream);
        }
        default: return VMV_EINVAL;
    }
}
size_t op_size(int op) {
    switch (op) {
        case VMV_OP_GEMM: return sizeof(VmvGemmParams);
        case VMV_OP_GN_STATS: case VMV_OP_GN_APPLY: case VMV_OP_GN_FUSED: case VMV_OP_GN_TABLE: return sizeof(VmvGroupNormParams);
        case VMV_OP_LAYERNORM: return sizeof(VmvLayerNormParams);
        case VMV_OP_ATTENTION: return sizeof(VmvAttnParams);
        case VMV_OP_SOFTMAX: return sizeof(VmvSoftmaxParams);
        case VMV_OP_COPY: return sizeof(VmvCopyParams);
        case VMV_OP_FF: return sizeof(VmvFfParams);
        case VMV_OP_COMM: return sizeof(VmvCommParams);
        default: return 0;
    }
}
}  // namespace

extern "C" int vmv_abi_version(void) { return VMV_ABI_VERSION; }
extern "C" int vmv_elem_type(void) { return VMV_ELEM_TYPE; }
extern "C" int vmv_sizeof(int which) {
    switch (which) {
        case 100: return (int)sizeof(VmvDdimParams);
        case 101: return (int)sizeof(VmvGemmSeg);
        case 102: return (int)sizeof(VmvSeqMap);
        case 103: return (int)sizeof(VmvGsParams);
        case 104: return (int)sizeof(VmvGsBatchParams);
        default: return (int)op_size(which);
    }
}

extern "C" const char* vmv_error_string(int code) {
    switch (code) {
        case VMV_OK: return "ok";
        case VMV_EINVAL: return "VMV_EINVAL: bad dimension or flag combination";
        case VMV_EALIGN: return "VMV_EALIGN: pointer or leading dimension not suitably aligned";
        case VMV_ENULL: return "VMV_ENULL: required pointer is NULL";
        case VMV_ERANGE: return "VMV_ERANGE: size outside what the kernel supports";
        case VMV_ECOMM: return "VMV_ECOMM: an RCCL call failed (vmv_comm_*)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown vmv error";
    }
}

extern "C" VmvPlan* vmv_plan_create(void) { return new (std::nothrow) VmvPlan(); }
extern "C" void vmv_plan_destroy(VmvPlan* plan) { delete plan; }
extern "C" int vmv_plan_add(VmvPlan* plan, int op, const void* params, size_t nbytes) {
    if (!plan || !params) return VMV_ENULL;
    const size_t want = op_size(op);
    if (want == 0 || nbytes != want) return VMV_EINVAL;
    VmvPlan::Op o;
    o.op = op;
    o.args.resize(nbytes);
    std::memcpy(o.args.data(), params, nbytes);
    plan->ops.push_back(std::move(o));
    return (int)plan->ops.size() - 1;
}
extern "C" int vmv_plan_size(const VmvPlan* plan) { return plan ? (int)plan->ops.size() : VMV_ENULL; }
extern "C" int vmv_plan_run_range(const VmvPlan* plan, int first, int last, void* stream) {
    if (!plan) return VMV_ENULL;
    if (first < 0 || last > (int)plan->ops.size() || first > last) return VMV_EINVAL;
    for (int i = first; i < last; ++i) {
        const int rc = run_op(plan->ops[i], stream);
        if (rc != VMV_OK) return rc;
    }
    return VMV_OK;
}
extern "C" int vmv_plan_run(const VmvPlan* plan, void* stream) {
    if (!plan) return VMV_ENULL;
    return vmv_plan_run_range(plan, 0, (int)plan->ops.size(), stream);
}

// ---- the plan as a hipGraph (vmv.h): one capture of a whole replay, instantiated once, launched per forward
struct VmvGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int nodes = 0;
};
extern "C" VmvGraph* vmv_plan_capture(const VmvPlan* plan, void* stream) {
    if (!plan || plan->ops.empty()) return nullptr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    VmvGraph* g = new (std::nothrow) VmvGraph();
    if (!g) return nullptr;
    // thread-local mode: other host threads (a watchdog, the data loader of a serving process) may keep making HIP calls
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { delete g; return nullptr; }
    int rc = VMV_OK;
    for (size_t i = 0; i < plan->ops.size() && rc == VMV_OK; ++i) rc = run_op(plan->ops[i], stream);
    const hipError_t e = hipStreamEndCapture(s, &g->graph);          // (always ends the capture, also after a failed launch)
    if (rc != VMV_OK || e != hipSuccess || !g->graph) { vmv_graph_destroy(g); return nullptr; }
    size_t n = 0;
    if (hipGraphGetNodes(g->graph, nullptr, &n) == hipSuccess) g->nodes = (int)n;
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) { vmv_graph_destroy(g); return nullptr; }
    return g;
}
extern "C" int vmv_graph_launch(const VmvGraph* g, void* stream) {
    if (!g || !g->exec) return VMV_ENULL;
    const hipError_t e = hipGraphLaunch(g->exec, reinterpret_cast<hipStream_t>(stream));
    return e == hipSuccess ? VMV_OK : (int)e;
}
extern "C" int vmv_graph_nodes(const VmvGraph* g) { return g ? g->nodes : VMV_ENULL; }
extern "C" void vmv_graph_destroy(VmvGraph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
