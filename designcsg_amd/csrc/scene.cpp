// scene.cpp -- parsers for scene.txt / buildprocedure.txt / exportConfig.txt / lookupTable.txt.
#include "scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "dcsg_device_params.h"

namespace dcsg {

static const char* skip_ws(const char* p)
{
    while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
    return p;
}

// DrawPane.cpp:280-353: per line "%d %d" then 12 x "%f", single-space separated.
// DrawPane.cpp:355-371: per line "%d %d %d %d".
bool parse_scene(const char* scene_txt, const char* bp_txt, Scene& s, std::string& err)
{
    s = Scene();
    if (!scene_txt || !bp_txt) {
        err = "scene_txt / buildprocedure_txt is NULL";
        return false;
    }
    const char* p = skip_ws(scene_txt);
    while (*p) {
        char* end;
        long brush = strtol(p, &end, 10);
        if (end == p) {
            err = "scene.txt: expected brush id at object " + std::to_string(s.n_objects);
            return false;
        }
        p = end;
        long material = strtol(p, &end, 10);  // material id: used by the preview only
        if (end == p) {
            err = "scene.txt: expected material id at object " + std::to_string(s.n_objects);
            return false;
        }
        p = end;
        float v[12];
        for (int k = 0; k < 12; k++) {
            v[k] = strtof(p, &end);
            if (end == p) {
                err = "scene.txt: object " + std::to_string(s.n_objects) + " has fewer than 12 floats";
                return false;
            }
            p = end;
        }
        s.shape.push_back((uint8_t)brush);
        s.material.push_back((uint8_t)material);
        for (int k = 0; k < 3; k++) {
            s.pos.push_back(v[k]);
            s.right.push_back(v[3 + k]);
            s.up.push_back(v[6 + k]);
            s.fwd.push_back(v[9 + k]);
        }
        s.n_objects++;
        p = skip_ws(p);
    }
    p = skip_ws(bp_txt);
    while (*p) {
        int v[4];
        for (int k = 0; k < 4; k++) {
            char* end;
            v[k] = (int)strtol(p, &end, 10);
            if (end == p) {
                err = "buildprocedure.txt: step " + std::to_string(s.n_steps) + " has fewer than 4 integers";
                return false;
            }
            p = end;
        }
        for (int k = 0; k < 4; k++) s.steps.push_back(v[k]);
        s.n_steps++;
        p = skip_ws(p);
    }
    // k2.cl's switch ignores opcodes it does not know (k2.cl:59-140): such steps are dropped HERE, before anything
    // looks at their operands.  The numbers 16.. and 32.. are the host's own fused / stack opcodes (optimise_program)
    // and must never be taken from a file.
    {
        std::vector<int32_t> kept;
        kept.reserve(s.steps.size());
        for (int c = 0; c < s.n_steps; c++)
            if (s.steps[4 * c] >= 0 && s.steps[4 * c] <= 8) kept.insert(kept.end(), s.steps.begin() + 4 * c, s.steps.begin() + 4 * c + 4);
        s.steps.swap(kept);
        s.n_steps = (int)(s.steps.size() / 4);
    }
    // Validation: everything the kernels index with must be in range.
    int max_slot = -1;
    auto slot_ok = [&](int a, const char* what, int step) {
        if (a < 0 || a >= DCSG_MAX_SLOTS) {
            err = std::string("buildprocedure.txt: step ") + std::to_string(step) + " " + what + " slot " +
                  std::to_string(a) + " outside 0.." + std::to_string(DCSG_MAX_SLOTS - 1) +
                  " (STACK_MEMORY_PER_PIXEL, Evaluator.cpp:7)";
            return false;
        }
        if (a > max_slot) max_slot = a;
        return true;
    };
    bool have_k = false;
    for (int c = 0; c < s.n_steps; c++) {
        int op = s.steps[4 * c], l = s.steps[4 * c + 1], r = s.steps[4 * c + 2], d = s.steps[4 * c + 3];
        switch (op) {
        case 0:  // IMPORT brush=l object=r -> d
            if (r < 0 || r >= s.n_objects) {
                err = "buildprocedure.txt: step " + std::to_string(c) + " imports object " + std::to_string(r) +
                      " but scene.txt has " + std::to_string(s.n_objects);
                return false;
            }
            if (!slot_ok(d, "destination", c)) return false;
            if ((l & 0xff) > s.max_brush_id) s.max_brush_id = l & 0xff;
            break;
        case 1:  // EXPORT l
            if (!slot_ok(l, "source", c)) return false;
            break;
        case 2:
        case 3:  // MIN / MAX l r -> d
            if (!slot_ok(l, "left", c) || !slot_ok(r, "right", c) || !slot_ok(d, "destination", c)) return false;
            break;
        case 4:
        case 5:  // NEGATE / IDENTITY l -> d
            if (!slot_ok(l, "source", c) || !slot_ok(d, "destination", c)) return false;
            break;
        case 6:
        case 7:  // extension: SMOOTH_MIN / SMOOTH_MAX l r -> d  (blend radius from the last SET_K)
            if (!slot_ok(l, "left", c) || !slot_ok(r, "right", c) || !slot_ok(d, "destination", c)) return false;
            if (!have_k) {
                err = "buildprocedure.txt: step " + std::to_string(c) + " is a smooth join but no SET_K step precedes it";
                return false;
            }
            s.has_smooth = true;
            break;
        case 8: {  // extension: SET_K bits(l)
            float k;
            int32_t bits = l;
            memcpy(&k, &bits, 4);
            if (!(k > 0.0f) || k - k != 0.0f) {
                err = "buildprocedure.txt: step " + std::to_string(c) + " sets a blend radius that is not finite and > 0";
                return false;
            }
            have_k = true;
        } break;
        default:  // unreachable: unknown opcodes were dropped above
            err = "internal: unknown opcode survived the filter";
            return false;
        }
    }
    s.n_slots = max_slot + 1;
    if (s.n_slots < 1) s.n_slots = 1;
    optimise_program(s);
    if (s.invalid_program) {
        err = "buildprocedure.txt: a step addresses a slot outside 0.." + std::to_string(DCSG_MAX_SLOTS - 1);
        return false;
    }
    return true;
}

size_t Scene::lds_bytes() const
{
    // blob + generic interpreter stack + the culling kernels' work area: per step LB, UB (f32), list entry (u16),
    // opcode (u8) and a counter -- also large enough for the 4 per-wave u16 lists of the projection kernel
    // (block list: 2 bytes per step + 80 bytes of wave totals + 4 per item for the lower bounds of a plain union: the 80 spare
    // bytes keep that inside the area for the shortest programs too)
    size_t lists = (((size_t)n_stack_prog + 7) & ~(size_t)7) * 12 + 16 + 80;
    return blob.size() * 4 + (size_t)n_regs * DCSG_BLOCK * sizeof(float) + ((lists + 15) & ~(size_t)15);
}

namespace {
struct Step {
    int op, l, r, d;
    uint32_t k = 0;  // blend radius (float bits) of a smooth join, else 0
};
enum {
    IMPORT = 0, EXPORT = 1, MIN = 2, MAX = 3, NEGATE = 4, IDENTITY = 5,
    SMIN = 6, SMAX = 7, SETK = 8,  // extension opcodes of the file format (smooth joins)
    IMPORT_MIN = 16, IMPORT_MAX = 17, IMPORT_NEGMAX = 18, IMPORT_SMIN = 19, IMPORT_SMAX = 20, IMPORT_NEGSMAX = 21
};

// {op, k bits, bits of 1.0f/k, 0}: the step that loads the blend radius before a smooth join whose radius differs
// from the one in force
void push_setk(std::vector<int32_t>& prog, int op, uint32_t kbits)
{
    float k, inv;
    memcpy(&k, &kbits, 4);
    inv = 1.0f / k;
    uint32_t ib;
    memcpy(&ib, &inv, 4);
    prog.insert(prog.end(), {op, (int32_t)kbits, (int32_t)ib, 0});
}

// slots a step reads / the slot it writes (-1: none)
void step_access(const Step& t, int reads[2], int& write)
{
    reads[0] = reads[1] = write = -1;
    switch (t.op) {
    case IMPORT: write = t.d; break;
    case IMPORT_MIN:
    case IMPORT_MAX:
    case IMPORT_NEGMAX:
    case IMPORT_SMIN:
    case IMPORT_SMAX:
    case IMPORT_NEGSMAX: reads[0] = t.d; write = t.d; break;
    case EXPORT: reads[0] = t.l; break;
    case MIN:
    case MAX:
    case SMIN:
    case SMAX: reads[0] = t.l; reads[1] = t.r; write = t.d; break;
    case NEGATE:
    case IDENTITY: reads[0] = t.l; write = t.d; break;
    default: break;
    }
}

// true when slot R is not read after step j before being overwritten (or the program ends)
bool dead_after(const std::vector<Step>& p, size_t j, int R)
{
    for (size_t k = j + 1; k < p.size(); k++) {
        int rd[2], wr;
        step_access(p[k], rd, wr);
        if (rd[0] == R || rd[1] == R) return false;
        if (wr == R) return true;
    }
    return true;
}
}  // namespace

void optimise_program(Scene& s)
{
    std::vector<Step> in, out;
    {
        uint32_t cur_k = 0;
        for (int c = 0; c < s.n_steps; c++) {
            Step t{s.steps[4 * c], s.steps[4 * c + 1], s.steps[4 * c + 2], s.steps[4 * c + 3]};
            if (t.op == SETK) {  // folded into the smooth joins it governs; re-materialised on emission
                cur_k = (uint32_t)t.l;
                continue;
            }
            if (t.op == SMIN || t.op == SMAX) t.k = cur_k;
            in.push_back(t);
        }
    }
    // ---- fusion ------------------------------------------------------------------------------
    for (size_t i = 0; i < in.size(); i++) {
        const Step& a = in[i];
        if (a.op == IMPORT && i + 1 < in.size()) {
            int R = a.d;
            const Step& b = in[i + 1];
            // IMPORT -> R ; MIN|MAX x R -> x      (scenecompiler.py:372,377)
            if ((b.op == MIN || b.op == MAX || b.op == SMIN || b.op == SMAX) && b.r == R && b.l == b.d && b.l != R &&
                dead_after(in, i + 1, R)) {
                int f = b.op == MIN ? IMPORT_MIN : (b.op == MAX ? IMPORT_MAX : (b.op == SMIN ? IMPORT_SMIN : IMPORT_SMAX));
                out.push_back({f, a.l, a.r, b.d, b.k});
                i += 1;
                continue;
            }
            // IMPORT -> R ; NEGATE R -> R ; MAX x R -> x   (scenecompiler.py:372-375)
            if (i + 2 < in.size() && b.op == NEGATE && b.l == R && b.d == R) {
                const Step& m = in[i + 2];
                if ((m.op == MAX || m.op == SMAX) && m.r == R && m.l == m.d && m.l != R && dead_after(in, i + 2, R)) {
                    out.push_back({m.op == MAX ? IMPORT_NEGMAX : IMPORT_NEGSMAX, a.l, a.r, m.d, m.k});
                    i += 2;
                    continue;
                }
            }
        }
        out.push_back(a);
    }
    // ---- stack program ----------------------------------------------------------------------------
    // The scene compiler's programs are stack-disciplined (scenecompiler.py:353-387: a group imports into
    // its own slot, accumulates its leaves, recurses into sub-groups and is joined into its parent once).
    // Replaying the register program on a stack of slot names converts it; any step that does not fit
    // leaves stack_prog empty and the generic interpreter runs the register program instead.
    {
        enum {
            PUSH = 32, ACC_MIN = 33, ACC_MAX = 34, ACC_NEGMAX = 35, POP_MIN = 36, POP_MAX = 37, POP_NEGMAX = 38, EXPORT_TOP = 39,
            S_SETK = 40, ACC_SMIN = 41, ACC_SMAX = 42, ACC_NEGSMAX = 43, POP_SMIN = 44, POP_SMAX = 45, POP_NEGSMAX = 46
        };
        std::vector<int> names;  // slot names, back() = top
        std::vector<Step> sp;
        bool ok = true;
        int depth_max = 0;
        auto on_stack = [&](int v) { return std::find(names.begin(), names.end(), v) != names.end(); };
        for (size_t i = 0; i < out.size() && ok; i++) {
            const Step& t = out[i];
            int n = (int)names.size();
            switch (t.op) {
            case IMPORT:
                if (on_stack(t.d)) { ok = false; break; }
                sp.push_back({PUSH, t.l, t.r, n});
                names.push_back(t.d);
                break;
            case IMPORT_MIN:
            case IMPORT_MAX:
            case IMPORT_NEGMAX:
                if (n < 1 || names.back() != t.d) { ok = false; break; }
                sp.push_back({t.op == IMPORT_MIN ? ACC_MIN : (t.op == IMPORT_MAX ? ACC_MAX : ACC_NEGMAX), t.l, t.r, n});
                break;
            case IMPORT_SMIN:
            case IMPORT_SMAX:
            case IMPORT_NEGSMAX:
                if (n < 1 || names.back() != t.d) { ok = false; break; }
                sp.push_back({t.op == IMPORT_SMIN ? ACC_SMIN : (t.op == IMPORT_SMAX ? ACC_SMAX : ACC_NEGSMAX), t.l, t.r, n, t.k});
                break;
            case MIN:
            case MAX:  // joinMode parent child -> parent  (scenecompiler.py:384)
            case SMIN:
            case SMAX:
                if (n < 2 || t.d != t.l || names[n - 2] != t.l || names[n - 1] != t.r || !dead_after(out, i, t.r)) { ok = false; break; }
                sp.push_back({t.op == MIN ? POP_MIN : (t.op == MAX ? POP_MAX : (t.op == SMIN ? POP_SMIN : POP_SMAX)), 0, 0, n, t.k});
                names.pop_back();
                break;
            case NEGATE: {  // NEGATE child -> R0 ; MAX parent R0 -> parent  (scenecompiler.py:381-382)
                if (i + 1 >= out.size() || n < 2) { ok = false; break; }
                const Step& m = out[i + 1];
                int R = t.d;
                if (names[n - 1] != t.l || on_stack(R) || (m.op != MAX && m.op != SMAX) || m.r != R || m.l != m.d ||
                    names[n - 2] != m.l || !dead_after(out, i + 1, R) || !dead_after(out, i + 1, t.l)) { ok = false; break; }
                sp.push_back({m.op == MAX ? POP_NEGMAX : POP_NEGSMAX, 0, 0, n, m.k});
                names.pop_back();
                i += 1;
            } break;
            case EXPORT:
                if (n < 1 || names.back() != t.l) { ok = false; break; }
                sp.push_back({EXPORT_TOP, 0, 0, n});
                break;
            case IDENTITY: ok = false; break;
            default: break;  // unknown opcodes are ignored by k2.cl's switch
            }
            if ((int)names.size() > depth_max) depth_max = (int)names.size();
        }
        s.stack_prog.clear();
        s.n_stack_prog = 0;
        s.stack_depth = 0;
        if (ok) {
            uint32_t cur_k = 0;
            for (const Step& t : sp) {
                // every group loads the radius it needs itself (a PUSH forgets the one in force): the tile culling
                // drops whole groups, and a later group must not depend on a SET_K inside a dropped one
                if (t.op == PUSH) cur_k = 0;
                if (t.k && t.k != cur_k) push_setk(s.stack_prog, S_SETK, cur_k = t.k);
                s.stack_prog.push_back(t.op);
                s.stack_prog.push_back(t.l);
                s.stack_prog.push_back(t.r);
                s.stack_prog.push_back(t.d);
            }
            s.n_stack_prog = (int)(s.stack_prog.size() / 4);
            s.stack_depth = depth_max;
        }
        // items: what the root accumulator sees -- single accumulate steps and whole groups (indices into the
        // emitted stack_prog, SET_K steps included).  Smooth joins are allowed INSIDE a group (its steps are walked
        // with interval arithmetic, kernels.hip dcsg_item_clamp); the root's own joins must be the hard ones, whose
        // action on an interval is a clamp.
        s.items.clear();
        s.n_items = 0;
        if (ok && depth_max <= 2) {
            bool fine = true;
            const std::vector<int32_t>& q = s.stack_prog;
            const size_t nq = q.size() / 4;
            auto OP = [&](size_t k) { return q[4 * k]; };
            auto DEPTH = [&](size_t k) { return q[4 * k + 3]; };
            for (size_t k = 0; k < nq && fine;) {
                int op = OP(k);
                if (op == PUSH && DEPTH(k) == 0) {
                    s.items.insert(s.items.end(), {(int)k, 1, PUSH, 0});
                    k++;
                } else if (op >= ACC_MIN && op <= ACC_NEGMAX && DEPTH(k) == 1) {
                    s.items.insert(s.items.end(), {(int)k, 1, op, 0});
                    k++;
                } else if (op == PUSH && DEPTH(k) == 1) {
                    size_t e = k + 1;
                    while (e < nq && !((OP(e) >= POP_MIN && OP(e) <= POP_NEGMAX) || (OP(e) >= POP_SMIN && OP(e) <= POP_NEGSMAX))) {
                        bool member = (OP(e) >= ACC_MIN && OP(e) <= ACC_NEGMAX) || (OP(e) >= ACC_SMIN && OP(e) <= ACC_NEGSMAX) ||
                                      OP(e) == S_SETK;
                        if (!member) fine = false;
                        e++;
                    }
                    if (e >= nq || OP(e) > POP_NEGMAX) fine = false;  // a smooth join into the root is not a clamp
                    if (fine) s.items.insert(s.items.end(), {(int)k, (int)(e - k + 1), OP(e), 0});
                    k = e + 1;
                } else if (op == EXPORT_TOP) {
                    s.items.insert(s.items.end(), {(int)k, 1, EXPORT_TOP, 0});
                    k++;
                } else {
                    fine = false;  // incl. smooth accumulates or SET_K at root level
                }
            }
            if (fine) {
                s.n_items = (int)(s.items.size() / 4);
                // items[i].w: the item thread i bounds in the culling's order-free first pass of a plain union
                // (kernels.hip, dcsg_top_min_upper) -- the items sorted by brush, so that the lanes of a wave run
                // through ONE brush's code instead of all of them under exec masks
                std::vector<int> order(s.n_items);
                for (int i = 0; i < s.n_items; i++) order[i] = i;
                auto key = [&](int i) {
                    const int first = s.items[4 * i], count = s.items[4 * i + 1], op = s.items[4 * i + 2];
                    if (count != 1 || op == EXPORT_TOP) return 1 << 20;
                    return (int)(q[4 * (size_t)first + 1] & 0xff);  // the step's brush
                };
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
                bool top_min = s.n_items > 1;  // the same test as build_blob's flag bit 1
                for (int i = 0; i < s.n_items && top_min; i++) {
                    const int op = s.items[4 * i + 2];
                    top_min = (i == 0 && op == PUSH) || (i > 0 && (op == ACC_MIN || op == POP_MIN || op == EXPORT_TOP));
                }
                for (int i = 0; i < s.n_items; i++) {
                    if (top_min) {
                        s.items[4 * i + 3] = order[i];
                        continue;
                    }
                    // otherwise items[i].w: a group that is ONE fold -- 1: PUSH then T_min steps only, 2: PUSH then
                    // T_max / T_max(-s) steps only -- whose leaves the culling bounds over all of them (dcsg_item_clamp)
                    const int first = s.items[4 * i], count = s.items[4 * i + 1];
                    int fold = 0;
                    if (count > 2 && count <= 64 && OP((size_t)first) == PUSH) {
                        bool all_min = true, all_max = true;
                        for (int j = 1; j < count - 1; j++) {
                            const int op = OP((size_t)(first + j));
                            all_min = all_min && op == ACC_MIN;
                            all_max = all_max && (op == ACC_MAX || op == ACC_NEGMAX);
                        }
                        fold = all_min ? 1 : (all_max ? 2 : 0);
                    }
                    s.items[4 * i + 3] = fold;
                }
            } else {
                s.items.clear();
            }
        }
    }
    // ---- liveness renaming ---------------------------------------------------------------------
    const int NV = DCSG_MAX_SLOTS;
    for (const Step& t : out) {  // parse_scene validated every slot; a violation here is a host bug, not bad input
        int rd[2], wr;
        step_access(t, rd, wr);
        for (int v : {rd[0], rd[1], wr})
            if (v < -1 || v >= NV) {  // leave no program behind: the caller reports the scene as invalid
                s.prog.clear();
                s.stack_prog.clear();
                s.n_prog = s.n_stack_prog = 0;
                s.invalid_program = true;
                return;
            }
    }
    int first[NV], last[NV];
    bool read_first[NV];
    for (int v = 0; v < NV; v++) first[v] = last[v] = -2, read_first[v] = false;
    for (size_t k = 0; k < out.size(); k++) {
        int rd[2], wr;
        step_access(out[k], rd, wr);
        for (int q = 0; q < 2; q++)
            if (rd[q] >= 0) {
                if (first[rd[q]] == -2) first[rd[q]] = (int)k, read_first[rd[q]] = true;
                last[rd[q]] = (int)k;
            }
        if (wr >= 0) {
            if (first[wr] == -2) first[wr] = (int)k;
            last[wr] = (int)k;
        }
    }
    std::vector<int> order;
    for (int v = 0; v < NV; v++)
        if (first[v] != -2) {
            if (read_first[v]) first[v] = -1;  // read before any write: must own a zeroed register from the start
            order.push_back(v);
        }
    std::sort(order.begin(), order.end(), [&](int a, int b) { return first[a] != first[b] ? first[a] < first[b] : a < b; });
    int map[NV];
    for (int v = 0; v < NV; v++) map[v] = 0;
    std::vector<int> reg_end;  // per physical register: last step of its current tenant
    for (int v : order) {
        int pick = -1;
        for (size_t r = 0; r < reg_end.size(); r++)
            if (reg_end[r] < first[v] && !read_first[v]) {  // a recycled register is not zero any more
                pick = (int)r;
                break;
            }
        if (pick < 0) {
            pick = (int)reg_end.size();
            reg_end.push_back(last[v]);
        } else {
            reg_end[pick] = last[v];
        }
        map[v] = pick;
    }
    s.n_regs = reg_end.empty() ? 1 : (int)reg_end.size();
    s.prog.clear();
    uint32_t cur_k = 0;
    for (const Step& t : out) {
        Step u = t;
        if (t.k && t.k != cur_k) push_setk(s.prog, SETK, cur_k = t.k);
        switch (t.op) {
        case IMPORT:
        case IMPORT_MIN:
        case IMPORT_MAX:
        case IMPORT_NEGMAX:
        case IMPORT_SMIN:
        case IMPORT_SMAX:
        case IMPORT_NEGSMAX: u.d = map[t.d]; break;
        case EXPORT: u.l = map[t.l]; break;
        case MIN:
        case MAX:
        case SMIN:
        case SMAX: u.l = map[t.l]; u.r = map[t.r]; u.d = map[t.d]; break;
        case NEGATE:
        case IDENTITY: u.l = map[t.l]; u.d = map[t.d]; break;
        default: break;
        }
        s.prog.push_back(u.op);
        s.prog.push_back(u.l);
        s.prog.push_back(u.r);
        s.prog.push_back(u.d);
    }
    s.n_prog = (int)(s.prog.size() / 4);
}

void build_blob(Scene& s, bool null_brush_is_const)
{
    auto f2u = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        return u;
    };
    if (s.prog.empty() && s.n_steps) optimise_program(s);
    s.blob.clear();
    s.blob.push_back((uint32_t)s.n_prog);
    s.blob.push_back((uint32_t)s.n_objects);
    s.blob.push_back((uint32_t)s.n_regs);
    // flags: bit 0 = the null brush is a constant; bit 1 = the root accumulator sees nothing but PUSH (first), hard unions
    // (ACC_MIN, groups joined by POP_MIN) and the final EXPORT -- the tile culling may then bound it by the smallest upper
    // bound of ALL items, not only of those before a step (kernels.hip, dcsg_top_min_upper)
    bool top_min = s.n_items > 1;
    for (int i = 0; i < s.n_items && top_min; i++) {
        const int op = s.items[4 * i + 2];
        // kernels.hip: SOP_PUSH 32, SOP_ACC_MIN 33, SOP_POP_MIN 36, SOP_EXPORT_TOP 39 (the enum of optimise_program)
        top_min = (i == 0 && op == 32) || (i > 0 && (op == 33 || op == 36 || op == 39));
    }
    // bit 2: the bounds of all steps (2 floats each) fit into one wave's columns of the generic interpreter's LDS stack
    // (n_regs rows of 64 floats): the per-wave list builder computes them one step per lane (kernels.hip, DcsgStepCache)
    const bool wave_cache = s.n_items > 0 && 2 * (size_t)s.n_stack_prog <= (size_t)s.n_regs * 64;
    s.blob.push_back((null_brush_is_const ? 1u : 0u) | (top_min ? 2u : 0u) | (wave_cache ? 4u : 0u));
    s.blob.push_back((uint32_t)s.n_stack_prog);
    s.blob.push_back((uint32_t)s.stack_depth);
    s.blob.push_back((uint32_t)s.n_items);
    s.blob.push_back(0u);
    for (int c = 0; c < 4 * s.n_prog; c++) s.blob.push_back((uint32_t)s.prog[c]);
    for (int c = 0; c < 4 * s.n_stack_prog; c++) s.blob.push_back((uint32_t)s.stack_prog[c]);
    for (int c = 0; c < 4 * s.n_items; c++) s.blob.push_back((uint32_t)s.items[c]);
    for (int i = 0; i < s.n_objects; i++) {
        s.blob.push_back(f2u(s.pos[3 * i]));
        s.blob.push_back(f2u(s.pos[3 * i + 1]));
        s.blob.push_back(f2u(s.pos[3 * i + 2]));
        s.blob.push_back((uint32_t)s.shape[i] | ((uint32_t)s.material[i] << 8));  // brush id, material id
        // Lipschitz bound of v -> (dot(v-o,right), dot(v-o,up), dot(v-o,forward)), used by the tile culling only: the
        // spectral norm of the matrix M with those rows = sqrt of the largest eigenvalue of M^T M (closed form for a
        // symmetric 3 x 3 matrix, in double, rounded up by 1e-6), never more than the Frobenius norm rounds 1-2 used --
        // up to sqrt(3) smaller for evenly scaled objects, and every margin of the culling is proportional to it.
        const float* rows[3] = {&s.right[3 * i], &s.up[3 * i], &s.fwd[3 * i]};
        double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, fro = 0;
        for (int r = 0; r < 3; r++)
            for (int u = 0; u < 3; u++) {
                fro += (double)rows[r][u] * rows[r][u];
                for (int v = 0; v < 3; v++) a[u][v] += (double)rows[r][u] * (double)rows[r][v];
            }
        double top = fro;  // trace of M^T M: an upper bound of its largest eigenvalue, and the fallback
        const double p1 = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (p1 == 0.0) {
            top = std::max(a[0][0], std::max(a[1][1], a[2][2]));
        } else {
            const double q = (a[0][0] + a[1][1] + a[2][2]) / 3.0;
            const double p2 = (a[0][0] - q) * (a[0][0] - q) + (a[1][1] - q) * (a[1][1] - q) + (a[2][2] - q) * (a[2][2] - q) + 2.0 * p1;
            const double pp = std::sqrt(p2 / 6.0);
            if (pp > 0.0 && std::isfinite(pp)) {
                double b[3][3];
                for (int u = 0; u < 3; u++)
                    for (int v = 0; v < 3; v++) b[u][v] = (a[u][v] - (u == v ? q : 0.0)) / pp;
                double det = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]) - b[0][1] * (b[1][0] * b[2][2] - b[1][2] * b[2][0]) +
                             b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
                double rr = 0.5 * det;
                rr = rr < -1.0 ? -1.0 : (rr > 1.0 ? 1.0 : rr);
                const double e = q + 2.0 * pp * std::cos(std::acos(rr) / 3.0);
                if (std::isfinite(e) && e > 0.0 && e < top) top = e;
            }
        }
        if (!(top <= fro)) top = fro;  // NaN / overflow: the Frobenius bound
        float lip = (float)(std::sqrt(top) * (1.0 + 1e-6)) + 1e-30f;
        const std::vector<float>* axes[3] = {&s.right, &s.up, &s.fwd};
        int ai = 0;
        for (auto* a : axes) {
            s.blob.push_back(f2u((*a)[3 * i]));
            s.blob.push_back(f2u((*a)[3 * i + 1]));
            s.blob.push_back(f2u((*a)[3 * i + 2]));
            s.blob.push_back(ai++ == 0 ? f2u(lip) : 0u);
        }
    }
}

// ---- scene.cl inspection ---------------------------------------------------------------------
static std::string strip_ws(const std::string& t)
{
    std::string o;
    for (char ch : t)
        if (ch != ' ' && ch != '\t' && ch != '\r' && ch != '\n') o.push_back(ch);
    return o;
}

// Body of "float sd<N>( float3 v){ ... }" as emitted by Brush.__str__ (scenecompiler.py:234-241),
// whitespace removed; empty if not found.
static std::string brush_body(const std::string& cl, int n)
{
    std::string key = "float sd" + std::to_string(n) + "(";
    size_t p = cl.find(key);
    if (p == std::string::npos) return "";
    size_t open = cl.find('{', p);
    if (open == std::string::npos) return "";
    int depth = 0;
    for (size_t i = open; i < cl.size(); i++) {
        if (cl[i] == '{') depth++;
        if (cl[i] == '}' && --depth == 0) return strip_ws(cl.substr(open + 1, i - open - 1));
    }
    return "";
}

bool scene_cl_null_brush_is_const(const std::string& cl)
{
    if (brush_body(cl, 0) != "returnMAX_DISTANCE;") return false;
    // any preprocessor line naming MAX_DISTANCE could change what sd0 returns
    size_t p = 0;
    while ((p = cl.find('#', p)) != std::string::npos) {
        size_t e = cl.find('\n', p);
        std::string line = cl.substr(p, e == std::string::npos ? std::string::npos : e - p);
        if (line.find("MAX_DISTANCE") != std::string::npos) return false;
        if (e == std::string::npos) break;
        p = e;
    }
    return true;
}

bool scene_cl_is_builtin_only(const std::string& cl)
{
    static const char* want[5] = {
        "returnMAX_DISTANCE;", "return0.0;", "returnlength(v)-0.5;",
        "v=fabs(v);floatx=length((float2)(v.x,v.z));floaty=v.y;returnmax(x-0.5,y-0.5);",
        "v=fabs(v);returnmax(v.x-0.5,max(v.y-0.5,v.z-0.5));"};
    if (cl.find('#') != std::string::npos) return false;  // user defines / AD offsets: compile it
    for (int n = 0; n < 5; n++)
        if (brush_body(cl, n) != want[n]) return false;
    return true;
}

// ---- exportConfig.txt --------------------------------------------------------------------------
bool parse_export_config(const char* text, dcsg_export_config& c, std::string& err)
{
    if (!text) {
        err = "export_config_txt is NULL";
        return false;
    }
    // cms::tokenize(contents, '\n') then std::stof / std::stoi per line, DesignCSG.cpp:825-835
    std::vector<std::string> lines;
    const char* p = text;
    while (*p) {
        const char* e = strchr(p, '\n');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len) lines.emplace_back(p, len);
        if (!e) break;
        p = e + 1;
    }
    if (lines.size() < 9) {
        err = "exportConfig.txt: expected 9 lines, got " + std::to_string(lines.size());
        return false;
    }
    char* end;
    auto F = [&](int i, float& o) {
        o = strtof(lines[i].c_str(), &end);
        return end != lines[i].c_str();
    };
    auto I = [&](int i, int32_t& o) {
        o = (int32_t)strtol(lines[i].c_str(), &end, 10);
        return end != lines[i].c_str();
    };
    bool ok = F(0, c.bounding_box_half_diameter) && I(1, c.min_level) && I(2, c.max_level) && I(3, c.grid_level) &&
              F(4, c.complex_surface_threshold) && I(5, c.gradient_descent_steps) && I(6, c.cache_subdivision) &&
              I(7, c.queries_before_gc) && I(8, c.queries_before_free);
    if (!ok) err = "exportConfig.txt: malformed number";
    return ok;
}

// ---- lookupTable.txt ---------------------------------------------------------------------------
// getIndexTriangleStrip, geometry.hpp:228-248
static int triangle_strip(const std::vector<int>& loop, int out[][3])
{
    int cnt = 0;
    size_t first = 0;
    if (loop.size() % 2 == 1) {
        out[cnt][0] = loop[0];
        out[cnt][1] = loop[1];
        out[cnt][2] = loop.back();
        cnt++;
        first = 1;
    }
    int m = (int)(loop.size() - first);
    for (int A = 0; A < m / 2 - 1; A++) {
        int B = A + 1, D = m - 1 - A, C = D - 1;
        out[cnt][0] = loop[first + A];
        out[cnt][1] = loop[first + B];
        out[cnt][2] = loop[first + C];
        cnt++;
        out[cnt][0] = loop[first + C];
        out[cnt][1] = loop[first + D];
        out[cnt][2] = loop[first + A];
        cnt++;
    }
    return cnt;
}

bool parse_lookup_table(const char* text, std::vector<uint8_t>& table, int& total, std::string& err)
{
    table.assign(256 * 16, 0);
    total = 0;
    int line = 0;
    const char* p = text;
    while (*p && line < 256) {
        const char* e = strchr(p, '\n');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == 0) {  // the tokenizer drops empty lines
            if (!e) break;
            p = e + 1;
            continue;
        }
        std::string s(p, len);
        if (s != "\t") {  // "\t" = case without triangles, readLookupTable.hpp:53
            size_t i = 0;
            while (i <= s.size()) {
                size_t semi = s.find(';', i);
                std::string cyc = s.substr(i, semi == std::string::npos ? std::string::npos : semi - i);
                std::vector<int> loop;
                const char* q = cyc.c_str();
                while (*q) {
                    char* end;
                    long v = strtol(q, &end, 10);
                    if (end == q) {
                        q++;
                        continue;
                    }
                    if (v < 0 || v > 11) {
                        err = "lookup table line " + std::to_string(line) + ": edge id out of range";
                        return false;
                    }
                    loop.push_back((int)v);
                    q = end;
                }
                if (loop.size() >= 3) {
                    if (loop.size() > 12) {
                        err = "lookup table line " + std::to_string(line) + ": loop too long";
                        return false;
                    }
                    int st[12][3];
                    int c = triangle_strip(loop, st);
                    for (int k = 0; k < c; k++) {
                        uint8_t& n = table[line * 16];
                        if (n >= 5) {
                            err = "lookup table line " + std::to_string(line) + ": more than 5 triangles";
                            return false;
                        }
                        for (int v = 0; v < 3; v++) table[line * 16 + 1 + 3 * n + v] = (uint8_t)st[k][v];
                        n++;
                        total++;
                    }
                }
                if (semi == std::string::npos) break;
                i = semi + 1;
            }
        }
        line++;
        if (!e) break;
        p = e + 1;
    }
    if (line != 256) {
        err = "lookup table has " + std::to_string(line) + " cases, expected 256";
        return false;
    }
    return true;
}

// ---- angle threshold ---------------------------------------------------------------------------
static int32_t float_to_ordered(float f)
{
    int32_t i;
    memcpy(&i, &f, 4);
    return i < 0 ? (int32_t)(0x80000000u - (uint32_t)i) : i;
}
static float ordered_to_float(int32_t o)
{
    int32_t i = o < 0 ? (int32_t)(0x80000000u - (uint32_t)o) : o;
    float f;
    memcpy(&f, &i, 4);
    return f;
}

float angle_threshold_to_q_star(float threshold)
{
    auto over = [&](float q) { return (float)acos((double)q) > threshold; };  // monotone non-increasing in q
    if (!over(-1.0f)) return -2.0f;
    if (over(1.0f)) return 1.0f;
    int32_t lo = float_to_ordered(-1.0f), hi = float_to_ordered(1.0f);  // over(lo) true, over(hi) false
    while (hi - lo > 1) {
        int32_t mid = lo + (hi - lo) / 2;
        if (over(ordered_to_float(mid)))
            lo = mid;
        else
            hi = mid;
    }
    return ordered_to_float(lo);
}

uint64_t fnv1a64(const void* data, size_t n, uint64_t h)
{
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

}  // namespace dcsg
