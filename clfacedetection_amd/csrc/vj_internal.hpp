// Internal host-side declarations shared by the translation units of libvjhip.so.
#pragma once
#include "../../include/vj.h"
#include <string>
#include <vector>
#include <cstdarg>

struct vj_cascade {
    int32_t win_w = 0, win_h = 0;
    std::vector<vj_stage_desc> stages;
    std::vector<vj_tree_desc>  trees;
    std::vector<vj_node_desc>  nodes;
    std::vector<float>         alpha;
    std::string notice;   // license / provenance comment of the source XML
    uint64_t uid = 0;     // unique per loaded object; keys the env's plan cache
    uint64_t content_hash = 0;   // FNV-1a of the window size and the four arrays: names the cascade across loads and processes
};
namespace vj { void finish_cascade(vj_cascade* c); }   // uid + content hash, once the arrays are final

namespace vj {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// One accepted scale as the device sees it.
struct ScalePlan {
    vj_scale_info info;
    uint32_t table_first;  // index of this scale's first node record in the table
};

// Resolved stage program: which stage a window visits after passing / failing
// stage s (tempcv.cpp:834-861 flattened).  -1 = accept, -2 = reject.
struct StageProgram {
    std::vector<int32_t> on_pass, on_fail;
    std::vector<uint32_t> n_nodes;      // nodes per stage
    std::vector<uint32_t> n_rects;      // sum of n_rects over the stage's nodes
    std::vector<uint32_t> n_roots, n_root_rects;   // trees per stage and the rectangles of their root nodes (every entering window evaluates those)
    std::vector<uint32_t> first_node;   // flat node index of the stage's first node
};
enum { STAGE_ACCEPT = -1, STAGE_REJECT = -2 };

StageProgram build_stage_program(const vj_cascade& c);
// Sweep order of the stage graph (topological, rooted at stage 0); false when the links form a cycle.
bool stage_sweep_order(const StageProgram& prog, std::vector<uint32_t>* order);

// Enumerate scales exactly as clod.cpp:1198-1204 + setupScale (clod.cpp:371-415).
std::vector<vj_scale_info> plan_scales(const vj_cascade& c, int width, int height,
                                       const vj_params& p);

// 16-dword device node record (see DESIGN.md "Feature table").
struct alignas(16) NodeRec {
    uint32_t lt[3];      // byte offset of the rect's left-top corner from the window origin
    uint32_t dh[3];      // byte distance top row -> bottom row (rh * stride * 4)
    uint32_t dw01;       // dw0 | dw1 << 16   (byte distance left -> right, rw * 4)
    uint32_t dw2_flags;  // dw2 | flags << 16
    float    w[3];       // scaled weights (w[2] == 0 when the node has two rects)
    float    thr;
    uint32_t left, right; // f32 leaf value bits, or child node index (flags)
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be 64 bytes");
enum { NODE_LEFT_IS_NODE = 1, NODE_RIGHT_IS_NODE = 2, NODE_TREE_LAST = 4 };

// Fill recs[n_nodes] for one scale (precomputeKernelCascade, clod.cpp:529-578).
int build_node_table(const vj_cascade& c, int width, const vj_scale_info& s, NodeRec* recs);
int build_node_table_stride(const vj_cascade& c, uint32_t stride, const vj_scale_info& s, NodeRec* recs,
                            uint32_t deint_half);

}  // namespace vj
