// Cascade loading: OpenCV old-format Haar XML -> flat POD arrays, and the compact
// .vjc binary form shipped with the package.
//
// Replaces cvLoad(xml) (main.cpp:36).  The schema and the tree/alpha/stage-link
// conventions follow icvReadHaarClassifier (tempcv.cpp:1749-2089): leaf values are
// appended to alpha[] in encounter order and referenced as left/right = -index
// (tempcv.cpp:1994-1995, 2032-2033); child = first stage naming this one as parent
// (tempcv.cpp:2080-2083).  Decimal fields go decimal -> double -> float, as OpenCV's
// (float)fn->data.f does (tempcv.cpp:1932, 1958, 1995, 2054).
#include "vj_internal.hpp"

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace vj {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<uint64_t> g_uid{1};

void finish_cascade(vj_cascade* c) {
    c->uid = g_uid++;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* data, size_t n) {
        const unsigned char* b = (const unsigned char*)data;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    mix(&c->win_w, 4);
    mix(&c->win_h, 4);
    mix(c->stages.data(), c->stages.size() * sizeof(vj_stage_desc));
    mix(c->trees.data(), c->trees.size() * sizeof(vj_tree_desc));
    mix(c->nodes.data(), c->nodes.size() * sizeof(vj_node_desc));
    mix(c->alpha.data(), c->alpha.size() * sizeof(float));
    c->content_hash = h;
}

// ---------------------------------------------------------------- tiny XML DOM
struct XmlNode {
    std::string name;
    std::string text;  // concatenated character data directly inside this element
    std::vector<std::unique_ptr<XmlNode>> kids;
    const XmlNode* child(const char* n) const {
        for (auto& k : kids)
            if (k->name == n) return k.get();
        return nullptr;
    }
};

struct XmlParser {
    const char* p;
    const char* end;
    std::string first_comment;
    bool have_comment = false;
    std::string err;

    bool starts(const char* s) const {
        size_t n = strlen(s);
        return (size_t)(end - p) >= n && memcmp(p, s, n) == 0;
    }
    // Skips <?...?>, <!--...-->, <!DOCTYPE...>. Returns false on unterminated markup.
    bool skip_misc() {
        for (;;) {
            while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
            if (starts("<?")) {
                const char* q = (const char*)memmem(p, end - p, "?>", 2);
                if (!q) { err = "unterminated <?"; return false; }
                p = q + 2;
            } else if (starts("<!--")) {
                const char* q = (const char*)memmem(p + 4, end - p - 4, "-->", 3);
                if (!q) { err = "unterminated comment"; return false; }
                if (!have_comment) { first_comment.assign(p + 4, q); have_comment = true; }
                p = q + 3;
            } else if (starts("<!")) {
                const char* q = (const char*)memchr(p, '>', end - p);
                if (!q) { err = "unterminated <!"; return false; }
                p = q + 1;
            } else {
                return true;
            }
        }
    }
    std::unique_ptr<XmlNode> element() {
        if (p >= end || *p != '<') { err = "expected '<'"; return nullptr; }
        ++p;
        const char* n0 = p;
        while (p < end && *p != '>' && *p != '/' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') ++p;
        auto node = std::make_unique<XmlNode>();
        node->name.assign(n0, p);
        // attributes are not needed by the Haar schema; skip to '>' honouring quotes
        bool self_close = false;
        while (p < end && *p != '>') {
            if (*p == '"' || *p == '\'') {
                char q = *p++;
                while (p < end && *p != q) ++p;
                if (p < end) ++p;
            } else {
                self_close = (*p == '/');
                ++p;
            }
        }
        if (p >= end) { err = "unterminated tag"; return nullptr; }
        ++p;  // '>'
        if (self_close) return node;
        for (;;) {
            const char* t0 = p;
            while (p < end && *p != '<') ++p;
            node->text.append(t0, p);
            if (p >= end) { err = "unexpected end inside <" + node->name + ">"; return nullptr; }
            if (starts("<!--") || starts("<?")) {
                if (!skip_misc()) return nullptr;
                continue;
            }
            if (starts("</")) {
                p += 2;
                const char* c0 = p;
                while (p < end && *p != '>') ++p;
                std::string cname(c0, p);
                while (!cname.empty() && (cname.back() == ' ' || cname.back() == '\n')) cname.pop_back();
                if (p >= end || cname != node->name) { err = "mismatched </" + cname + ">"; return nullptr; }
                ++p;
                return node;
            }
            auto kid = element();
            if (!kid) return nullptr;
            node->kids.push_back(std::move(kid));
        }
    }
};

static bool parse_ints(const std::string& s, int* out, int n_expected) {
    const char* q = s.c_str();
    for (int i = 0; i < n_expected; ++i) {
        char* e;
        errno = 0;
        long v = strtol(q, &e, 10);
        if (e == q || errno) return false;
        out[i] = (int)v;
        q = e;
    }
    return true;
}
static bool parse_real(const std::string& s, float* out) {
    const char* q = s.c_str();
    char* e;
    double d = strtod(q, &e);  // decimal -> f64 ...
    if (e == q) return false;
    *out = (float)d;           // ... -> f32, OpenCV's route
    return true;
}
static bool parse_int1(const XmlNode* n, int* out) { return n && parse_ints(n->text, out, 1); }

static int fail_parse(const char* what, int st, int tr, int nd) {
    set_error("cascade XML: %s (stage %d, tree %d, node %d)", what, st, tr, nd);
    return VJ_ERR_PARSE;
}

static int from_xml(const XmlNode* root, vj_cascade* c) {
    const XmlNode* sz = root->child("size");
    int wh[2];
    if (!sz || !parse_ints(sz->text, wh, 2) || wh[0] <= 0 || wh[1] <= 0)
        return fail_parse("size node is not two positive integers", -1, -1, -1);
    c->win_w = wh[0];
    c->win_h = wh[1];
    const XmlNode* stages = root->child("stages");
    if (!stages || stages->kids.empty()) return fail_parse("invalid stages node", -1, -1, -1);
    const int n_stages = (int)stages->kids.size();
    for (int i = 0; i < n_stages; ++i) {
        const XmlNode* st = stages->kids[i].get();
        const XmlNode* trees = st->child("trees");
        if (!trees || trees->kids.empty()) return fail_parse("trees node is not a valid sequence", i, -1, -1);
        vj_stage_desc sd;
        sd.first_tree = (int)c->trees.size();
        sd.n_trees = (int)trees->kids.size();
        for (int j = 0; j < sd.n_trees; ++j) {
            const XmlNode* tree = trees->kids[j].get();
            const int n_nodes = (int)tree->kids.size();
            if (n_nodes <= 0) return fail_parse("tree node is not a valid sequence", i, j, -1);
            vj_tree_desc td;
            td.first_node = (int)c->nodes.size();
            td.n_nodes = n_nodes;
            td.first_alpha = (int)c->alpha.size();
            int last_idx = 0;
            for (int k = 0; k < n_nodes; ++k) {
                const XmlNode* nd = tree->kids[k].get();
                const XmlNode* feat = nd->child("feature");
                const XmlNode* rects = feat ? feat->child("rects") : nullptr;
                if (!rects || rects->kids.empty() || rects->kids.size() > 3)
                    return fail_parse("rects node is not a valid sequence", i, j, k);
                vj_node_desc node;
                memset(&node, 0, sizeof(node));
                const int nr = (int)rects->kids.size();
                for (int l = 0; l < nr; ++l) {
                    // "x y w h weight" — four ints and a real
                    const std::string& t = rects->kids[l]->text;
                    int v[4];
                    if (!parse_ints(t, v, 4)) return fail_parse("rect is not a valid sequence", i, j, k);
                    // weight = 5th token
                    const char* q = t.c_str();
                    char* e;
                    for (int s = 0; s < 4; ++s) { strtol(q, &e, 10); q = e; }
                    float w;
                    if (!parse_real(q, &w)) return fail_parse("rect weight must be a real number", i, j, k);
                    if (v[0] < 0 || v[1] < 0 || v[2] <= 0 || v[3] <= 0 || v[0] + v[2] > c->win_w ||
                        v[1] + v[3] > c->win_h)
                        return fail_parse("rect exceeds the window", i, j, k);
                    node.rect[l] = {v[0], v[1], v[2], v[3], w};
                }
                int n_eff = 0;
                for (int l = 0; l < 3; ++l)
                    if (node.rect[l].weight != 0.0f) n_eff = l + 1;
                node.n_rects = n_eff;
                int tilted;
                if (!parse_int1(feat->child("tilted"), &tilted)) return fail_parse("tilted must be 0 or 1", i, j, k);
                node.tilted = tilted != 0;
                const XmlNode* thr = nd->child("threshold");
                if (!thr || !parse_real(thr->text, &node.threshold))
                    return fail_parse("threshold must be a real number", i, j, k);
                // left
                if (const XmlNode* ln = nd->child("left_node")) {
                    int v;
                    if (!parse_int1(ln, &v) || v <= k || v >= n_nodes)
                        return fail_parse("left node must be a valid node number", i, j, k);
                    node.left = v;
                } else {
                    const XmlNode* lv = nd->child("left_val");
                    float f;
                    if (!lv || !parse_real(lv->text, &f))
                        return fail_parse("left node or left value must be specified", i, j, k);
                    if (last_idx >= n_nodes + 1) return fail_parse("tree structure is broken: too many values", i, j, k);
                    node.left = -last_idx;
                    c->alpha.push_back(f);
                    ++last_idx;
                }
                // right
                if (const XmlNode* rn = nd->child("right_node")) {
                    int v;
                    if (!parse_int1(rn, &v) || v <= k || v >= n_nodes)
                        return fail_parse("right node must be a valid node number", i, j, k);
                    node.right = v;
                } else {
                    const XmlNode* rv = nd->child("right_val");
                    float f;
                    if (!rv || !parse_real(rv->text, &f))
                        return fail_parse("right node or right value must be specified", i, j, k);
                    if (last_idx >= n_nodes + 1) return fail_parse("tree structure is broken: too many values", i, j, k);
                    node.right = -last_idx;
                    c->alpha.push_back(f);
                    ++last_idx;
                }
                c->nodes.push_back(node);
            }
            if (last_idx != n_nodes + 1) return fail_parse("tree structure is broken: too few values", i, j, -1);
            c->trees.push_back(td);
        }
        const XmlNode* sthr = st->child("stage_threshold");
        if (!sthr || !parse_real(sthr->text, &sd.threshold))
            return fail_parse("stage threshold must be a real number", i, -1, -1);
        int parent, next;
        if (!parse_int1(st->child("parent"), &parent) || parent < -1 || parent >= n_stages)
            return fail_parse("parent must be an integer number", i, -1, -1);
        if (!parse_int1(st->child("next"), &next) || next < -1 || next >= n_stages)
            return fail_parse("next must be an integer number", i, -1, -1);
        sd.parent = parent;
        sd.next = next;
        sd.child = -1;
        c->stages.push_back(sd);
        if (parent != -1) {
            if (parent >= i) return fail_parse("parent must precede its child", i, -1, -1);
            if (c->stages[parent].child == -1) c->stages[parent].child = i;
        }
    }
    return VJ_OK;
}

static int read_file(const char* path, std::string* out) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error("cannot open %s: %s", path, strerror(errno));
        return VJ_ERR_IO;
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(&(*out)[0], 1, (size_t)n, f) : 0;
    fclose(f);
    if ((long)got != n) {
        set_error("short read on %s", path);
        return VJ_ERR_IO;
    }
    return VJ_OK;
}

StageProgram build_stage_program(const vj_cascade& c) {
    const int n = (int)c.stages.size();
    StageProgram sp;
    sp.on_pass.resize(n);
    sp.on_fail.resize(n);
    sp.n_nodes.assign(n, 0);
    sp.n_rects.assign(n, 0);
    sp.n_roots.assign(n, 0);
    sp.n_root_rects.assign(n, 0);
    sp.first_node.assign(n, 0);
    for (int s = 0; s < n; ++s) {
        const vj_stage_desc& sd = c.stages[s];
        // pass: ptr = ptr->child; NULL child ends the walk with "accept" (tempcv.cpp:849-852, 971)
        sp.on_pass[s] = sd.child >= 0 ? sd.child : STAGE_ACCEPT;
        // fail: climb while there is no next sibling (tempcv.cpp:855-858)
        int ptr = s;
        while (ptr != -1 && c.stages[ptr].next == -1) ptr = c.stages[ptr].parent;
        sp.on_fail[s] = ptr == -1 ? STAGE_REJECT : c.stages[ptr].next;
        sp.first_node[s] = (uint32_t)c.trees[sd.first_tree].first_node;
        for (int t = 0; t < sd.n_trees; ++t) {
            const vj_tree_desc& td = c.trees[sd.first_tree + t];
            sp.n_nodes[s] += (uint32_t)td.n_nodes;
            sp.n_roots[s] += 1u;
            sp.n_root_rects[s] += (uint32_t)c.nodes[td.first_node].n_rects;
            for (int k = 0; k < td.n_nodes; ++k) sp.n_rects[s] += (uint32_t)c.nodes[td.first_node + k].n_rects;
        }
    }
    return sp;
}

// Topological order of the pass / fail graph rooted at stage 0 (depth-first, pass edge first; reverse post-order):
// a window's walk through the stage tree only ever moves forward in it.  Unreachable stages are dropped.
bool stage_sweep_order(const StageProgram& prog, std::vector<uint32_t>* order) {
    const int nS = (int)prog.on_pass.size();
    std::vector<int> state(nS, 0);  // 0 unvisited, 1 on stack, 2 done
    std::vector<uint32_t> post;
    bool cyclic = false;
    std::vector<std::pair<int, int>> stack{{0, 0}};
    state[0] = 1;
    while (!stack.empty()) {
        auto& [s, phase] = stack.back();
        if (phase < 2) {
            const int nxt = phase == 0 ? prog.on_pass[s] : prog.on_fail[s];
            ++phase;
            if (nxt >= 0) {
                if (state[nxt] == 1) cyclic = true;
                if (state[nxt] == 0) {
                    state[nxt] = 1;
                    stack.push_back({nxt, 0});
                }
            }
        } else {
            state[s] = 2;
            post.push_back((uint32_t)s);
            stack.pop_back();
        }
    }
    order->assign(post.rbegin(), post.rend());
    return !cyclic;
}

}  // namespace vj

// ------------------------------------------------------------------- C ABI
using namespace vj;

extern "C" {

const char* vj_last_error(void) { return g_err; }

const char* vj_strerror(int code) {
    switch (code) {
        case VJ_OK: return "ok";
        case VJ_ERR_ARG: return "invalid argument";
        case VJ_ERR_IO: return "i/o error";
        case VJ_ERR_PARSE: return "malformed cascade file";
        case VJ_ERR_UNSUPPORTED: return "unsupported cascade feature";
        case VJ_ERR_NO_DEVICE: return "no usable HIP device";
        case VJ_ERR_HIP: return "HIP runtime error";
        case VJ_ERR_NOMEM: return "out of memory";
        case VJ_ERR_LIMIT: return "addressing limit exceeded";
        default: return "unknown error";
    }
}

int vj_cascade_load_xml(const char* path, vj_cascade** out) {
    if (!path || !out) return VJ_ERR_ARG;
    *out = nullptr;
    std::string buf;
    int rc = read_file(path, &buf);
    if (rc) return rc;
    XmlParser xp{buf.data(), buf.data() + buf.size(), std::string(), false, std::string()};
    if (!xp.skip_misc()) { set_error("%s: %s", path, xp.err.c_str()); return VJ_ERR_PARSE; }
    auto root = xp.element();
    if (!root) { set_error("%s: %s", path, xp.err.c_str()); return VJ_ERR_PARSE; }
    if (root->name != "opencv_storage" || root->kids.empty()) {
        set_error("%s: not an <opencv_storage> document", path);
        return VJ_ERR_PARSE;
    }
    auto c = std::make_unique<vj_cascade>();
    rc = from_xml(root->kids[0].get(), c.get());
    if (rc) return rc;
    c->notice = "Converted from OpenCV " + root->kids[0]->name + ".xml. Original notice:\n" + xp.first_comment;
    vj::finish_cascade(c.get());
    *out = c.release();
    return VJ_OK;
}

// .vjc layout (little endian):
//   char magic[8] "VJCASC01"; u32 notice_len; char notice[notice_len padded to 4];
//   i32 win_w, win_h, n_stages, n_trees, n_nodes, n_alpha;
//   vj_stage_desc[n_stages]; vj_tree_desc[n_trees]; vj_node_desc[n_nodes]; f32[n_alpha]
// Structural validation shared by every way a cascade gets in (indices are trusted by the table builders).
static int validate_cascade(const vj_cascade& c, const char* what) {
    const int nS = (int)c.stages.size(), nT = (int)c.trees.size(), nN = (int)c.nodes.size(), nA = (int)c.alpha.size();
    if (c.win_w <= 0 || c.win_h <= 0 || c.win_w > 4096 || c.win_h > 4096 || nS <= 0 || nT <= 0 || nN <= 0 || nA <= 0) {
        set_error("%s: empty cascade or bad window size", what);
        return VJ_ERR_PARSE;
    }
    for (int i = 0; i < nS; ++i) {
        const vj_stage_desc& s = c.stages[i];
        // a parent precedes its child (icvReadHaarClassifier reads stages in order; the fail walk climbs parents)
        if (s.first_tree < 0 || s.n_trees <= 0 || s.first_tree > nT - s.n_trees || s.parent < -1 || s.parent >= i ||
            s.next < -1 || s.next >= nS || s.child < -1 || s.child >= nS) {
            set_error("%s: stage links out of range", what);
            return VJ_ERR_PARSE;
        }
    }
    for (const auto& t : c.trees) {
        if (t.first_node < 0 || t.n_nodes <= 0 || t.first_node > nN - t.n_nodes || t.first_alpha < 0 ||
            t.n_nodes + 1 > nA || t.first_alpha > nA - t.n_nodes - 1) {
            set_error("%s: tree links out of range", what);
            return VJ_ERR_PARSE;
        }
        for (int k = 0; k < t.n_nodes; ++k) {
            const vj_node_desc& n = c.nodes[t.first_node + k];
            if (n.left >= t.n_nodes || n.right >= t.n_nodes || n.left < -t.n_nodes || n.right < -t.n_nodes ||
                (n.left > 0 && n.left <= k) || (n.right > 0 && n.right <= k) || n.n_rects < 1 || n.n_rects > 3) {
                set_error("%s: node links out of range", what);
                return VJ_ERR_PARSE;
            }
            for (int q = 0; q < 3; ++q) {
                const vj_rect_desc& r = n.rect[q];
                if (r.x < 0 || r.y < 0 || r.w < 0 || r.h < 0 || r.x > 4096 || r.y > 4096 || r.w > 4096 || r.h > 4096 ||
                    !(r.weight == r.weight)) {
                    set_error("%s: node %d rect %d out of range", what, t.first_node + k, q);
                    return VJ_ERR_PARSE;
                }
            }
        }
    }
    return VJ_OK;
}

static const char kMagic[8] = {'V', 'J', 'C', 'A', 'S', 'C', '0', '1'};

int vj_cascade_save(const vj_cascade* c, const char* path) {
    if (!c || !path) return VJ_ERR_ARG;
    FILE* f = fopen(path, "wb");
    if (!f) { set_error("cannot create %s: %s", path, strerror(errno)); return VJ_ERR_IO; }
    bool ok = fwrite(kMagic, 1, 8, f) == 8;
    uint32_t nl = (uint32_t)c->notice.size();
    uint32_t nl_pad = (nl + 3u) & ~3u;
    std::string notice = c->notice;
    notice.resize(nl_pad, '\n');
    ok = ok && fwrite(&nl_pad, 4, 1, f) == 1 && (nl_pad == 0 || fwrite(notice.data(), 1, nl_pad, f) == nl_pad);
    int32_t hdr[6] = {c->win_w, c->win_h, (int32_t)c->stages.size(), (int32_t)c->trees.size(),
                      (int32_t)c->nodes.size(), (int32_t)c->alpha.size()};
    ok = ok && fwrite(hdr, 4, 6, f) == 6;
    ok = ok && fwrite(c->stages.data(), sizeof(vj_stage_desc), c->stages.size(), f) == c->stages.size();
    ok = ok && fwrite(c->trees.data(), sizeof(vj_tree_desc), c->trees.size(), f) == c->trees.size();
    ok = ok && fwrite(c->nodes.data(), sizeof(vj_node_desc), c->nodes.size(), f) == c->nodes.size();
    ok = ok && fwrite(c->alpha.data(), sizeof(float), c->alpha.size(), f) == c->alpha.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("write error on %s", path); return VJ_ERR_IO; }
    return VJ_OK;
}

int vj_cascade_load(const char* path, vj_cascade** out) {
    if (!path || !out) return VJ_ERR_ARG;
    *out = nullptr;
    std::string buf;
    int rc = read_file(path, &buf);
    if (rc) return rc;
    const char* p = buf.data();
    const char* end = p + buf.size();
    auto need = [&](size_t n) { return (size_t)(end - p) >= n; };
    if (!need(12) || memcmp(p, kMagic, 8) != 0) { set_error("%s: not a VJCASC01 file", path); return VJ_ERR_PARSE; }
    p += 8;
    uint32_t nl;
    memcpy(&nl, p, 4);
    p += 4;
    if (!need((size_t)nl + 24)) { set_error("%s: truncated header", path); return VJ_ERR_PARSE; }
    auto c = std::make_unique<vj_cascade>();
    c->notice.assign(p, nl);
    p += nl;
    int32_t hdr[6];
    memcpy(hdr, p, 24);
    p += 24;
    for (int i = 0; i < 6; ++i)
        if (hdr[i] <= 0 || hdr[i] > (1 << 24)) { set_error("%s: bad header field %d", path, i); return VJ_ERR_PARSE; }
    c->win_w = hdr[0];
    c->win_h = hdr[1];
    size_t bytes = (size_t)hdr[2] * sizeof(vj_stage_desc) + (size_t)hdr[3] * sizeof(vj_tree_desc) +
                   (size_t)hdr[4] * sizeof(vj_node_desc) + (size_t)hdr[5] * sizeof(float);
    if ((size_t)(end - p) != bytes) { set_error("%s: payload size mismatch", path); return VJ_ERR_PARSE; }
    c->stages.resize(hdr[2]);
    c->trees.resize(hdr[3]);
    c->nodes.resize(hdr[4]);
    c->alpha.resize(hdr[5]);
    memcpy(c->stages.data(), p, c->stages.size() * sizeof(vj_stage_desc));
    p += c->stages.size() * sizeof(vj_stage_desc);
    memcpy(c->trees.data(), p, c->trees.size() * sizeof(vj_tree_desc));
    p += c->trees.size() * sizeof(vj_tree_desc);
    memcpy(c->nodes.data(), p, c->nodes.size() * sizeof(vj_node_desc));
    p += c->nodes.size() * sizeof(vj_node_desc);
    memcpy(c->alpha.data(), p, c->alpha.size() * sizeof(float));
    if ((rc = validate_cascade(*c, path))) return rc;
    vj::finish_cascade(c.get());
    *out = c.release();
    return VJ_OK;
}

void vj_cascade_free(vj_cascade* c) { delete c; }

// A cascade the caller already holds in memory (the reference's caller has a CvHaarClassifierCascade* from
// cvLoad, main.cpp:36; layout tempcv.hpp:70-112): the arrays are copied and validated like a file's.
int vj_cascade_from_arrays(int win_w, int win_h, const vj_stage_desc* stages, int n_stages, const vj_tree_desc* trees,
                           int n_trees, const vj_node_desc* nodes, int n_nodes, const float* alpha, int n_alpha,
                           vj_cascade** out) {
    if (!out) return VJ_ERR_ARG;
    *out = nullptr;
    if (!stages || !trees || !nodes || !alpha || n_stages <= 0 || n_trees <= 0 || n_nodes <= 0 || n_alpha <= 0 ||
        n_stages > (1 << 24) || n_trees > (1 << 24) || n_nodes > (1 << 24) || n_alpha > (1 << 24))
        return VJ_ERR_ARG;
    auto c = std::make_unique<vj_cascade>();
    c->win_w = win_w;
    c->win_h = win_h;
    c->stages.assign(stages, stages + n_stages);
    c->trees.assign(trees, trees + n_trees);
    c->nodes.assign(nodes, nodes + n_nodes);
    c->alpha.assign(alpha, alpha + n_alpha);
    // `child` as icvReadHaarClassifier derives it (tempcv.cpp:2080-2083) when the caller left it unset (all -1)
    bool any_child = false;
    for (const auto& s : c->stages) any_child |= s.child != -1;
    if (!any_child) {
        for (int i = 0; i < n_stages; ++i) {
            const int pa = c->stages[i].parent;
            if (pa >= 0 && pa < n_stages && c->stages[pa].child == -1) c->stages[pa].child = i;
        }
    }
    int rc = validate_cascade(*c, "vj_cascade_from_arrays");
    if (rc) return rc;
    vj::finish_cascade(c.get());
    *out = c.release();
    return VJ_OK;
}


int vj_cascade_get_info(const vj_cascade* c, vj_cascade_info* o) {
    if (!c || !o) return VJ_ERR_ARG;
    memset(o, 0, sizeof(*o));
    o->win_w = c->win_w;
    o->win_h = c->win_h;
    o->n_stages = (int)c->stages.size();
    o->n_trees = (int)c->trees.size();
    o->n_nodes = (int)c->nodes.size();
    o->n_alpha = (int)c->alpha.size();
    o->is_stump_based = 1;
    for (const auto& s : c->stages) {
        if (s.n_trees > o->max_trees_per_stage) o->max_trees_per_stage = s.n_trees;
        if (s.next != -1) o->is_stage_tree = 1;
    }
    for (const auto& t : c->trees) {
        if (t.n_nodes > o->max_nodes_per_tree) o->max_nodes_per_tree = t.n_nodes;
        if (t.n_nodes != 1) o->is_stump_based = 0;
    }
    for (const auto& n : c->nodes) {
        if (n.tilted) o->n_tilted++;
        if (n.n_rects == 3) o->n_three_rect++;
    }
    return VJ_OK;
}

const vj_stage_desc* vj_cascade_stages(const vj_cascade* c) { return c ? c->stages.data() : nullptr; }
const vj_tree_desc* vj_cascade_trees(const vj_cascade* c) { return c ? c->trees.data() : nullptr; }
const vj_node_desc* vj_cascade_nodes(const vj_cascade* c) { return c ? c->nodes.data() : nullptr; }
const float* vj_cascade_alpha(const vj_cascade* c) { return c ? c->alpha.data() : nullptr; }
const char* vj_cascade_notice(const vj_cascade* c) { return c ? c->notice.c_str() : nullptr; }

void vj_params_default(vj_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->scale_factor = 1.1f;
}

}  // extern "C"
