// The program of ONE degree class, compiled from the class's terms through a hash-consed expression graph (round 6).
//
// What halo2's GraphEvaluator does for a circuit like the EVM circuit (plonk/evaluation.rs, external crate: every sub-expression of
// every gate becomes ONE `Calculation`, whoever uses it) has to happen on THIS side of the boundary: the Rust exporter walks
// expression trees and emits them as they stand.  An EVM-style constraint system [REF zkevm-circuits/src/evm_circuit/execution.rs:832-851]
// is thousands of polynomials  q_usable * q_step * state_selector_s * (constraint * condition)  whose selector products are shared by
// every constraint of an execution state and whose conditions are shared by the constraints of a gadget
// [REF zkevm-circuits/src/evm_circuit/util/constraint_builder.rs:322-341].  Evaluated term by term that is two to four products per
// constraint spent on recomputing shared factors.
//
// compile_class():
//   1. every term becomes a node of one graph; equal sub-expressions (commutative operands ordered) are ONE node, whatever the
//      exporter parked or did not park (TEE_TMP / PUSH_TMP of the incoming programs are resolved into graph references);
//   2. the y-weighted sum of the terms is regrouped by common FACTORS, recursively:
//          sum_i y^(L-i) F g_i  =  F * sum_i y^(L-i) g_i           F = any node (a selector product, a condition), not only a column
//      so a state's selector product multiplies once per state and a gadget's condition once per gadget; every sum is evaluated in
//      Horner form over the constraint index (one product by a power of y per term, no weights table, no settling of partial sums);
//   3. nodes that are still used more than once and contain a product are parked; parking slots are assigned by LIVENESS (a slot is
//      free again after the last read of its value), so the parking area is as large as the number of values alive at once --
//      not as the number of shared sub-expressions of the circuit.
// The value is exact field arithmetic on the same polynomial: h and every proof byte are what they were.  The program computes
//      acc = sum_t y^(last - cons_t) term_t          (last = the largest constraint index of the class)
// and the caller scales by y^(K-1-last) as before.  ZK_QUOTIENT_DAG=0 keeps round 5's assembly (assemble_grouped / folding).
#pragma once

struct ClassCompileStats { uint32_t nodes = 0, parked = 0, max_live = 0, products = 0, instrs = 0, groups = 0; int depth = 0; };

struct ClassCompiler {
    enum : uint32_t { OP_HSUM = 0x100 };           // n-ary Horner sum (items in hs), output graph only
    struct Node { uint32_t op, a, b; int32_t x, y; };
    struct Item { int32_t node; uint32_t idx; };
    struct KeyHash {
        size_t operator()(const std::array<uint32_t, 5>& k) const {
            uint64_t h = 0x9E3779B97F4A7C15ull;
            for (uint32_t v : k) { h ^= v; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
            return (size_t)h;
        }
    };
    std::vector<Node> nd;
    std::vector<std::vector<Item>> hs;              // items of the HSUM nodes
    std::unordered_map<std::array<uint32_t, 5>, int32_t, KeyHash> index;
    std::vector<int8_t> prod_memo;                  // -1 unknown, 0 / 1: the subtree holds a product
    std::vector<uint32_t> cost_memo;                // products of the subtree, counted as a tree (what recomputing the node costs); 0 = unknown
    std::vector<uint8_t> no_park;                   // nodes that are recomputed at every use (parking them would hold a slot too long for what they cost)
    ClassCompileStats stats;

    int32_t intern(uint32_t op, uint32_t a, uint32_t b, int32_t x, int32_t y) {
        if ((op == Q_ADD || op == Q_MUL) && x > y) std::swap(x, y);
        const std::array<uint32_t, 5> key{op, a, b, (uint32_t)x, (uint32_t)y};
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        nd.push_back({op, a, b, x, y});
        index.emplace(key, (int32_t)nd.size() - 1);
        return (int32_t)nd.size() - 1;
    }
    int32_t fresh(uint32_t op, uint32_t a, uint32_t b, int32_t x, int32_t y) { nd.push_back({op, a, b, x, y}); return (int32_t)nd.size() - 1; }

    // postfix program -> graph node; slots = the parking area as the terms seen so far left it
    int32_t read(const Prog& g, std::unordered_map<uint32_t, int32_t>& slots) {
        std::vector<int32_t> st;
        for (const Instr& in : g) {
            switch (in.op) {
                case Q_PUSH_COL: st.push_back(intern(Q_PUSH_COL, in.a, in.b, -1, -1)); break;
                case Q_PUSH_CONST: st.push_back(intern(Q_PUSH_CONST, in.a, 0, -1, -1)); break;
                case Q_ADD: case Q_SUB: case Q_MUL: {
                    if (st.size() < 2) return -1;
                    const int32_t y = st.back(); st.pop_back();
                    st.back() = intern(in.op, 0, 0, st.back(), y);
                    break;
                }
                case Q_NEG: case Q_SQUARE: case Q_DOUBLE: if (st.empty()) return -1; st.back() = intern(in.op, 0, 0, st.back(), -1); break;
                case Q_MUL_CONST: case Q_ADD_CONST: if (st.empty()) return -1; st.back() = intern(in.op, in.a, 0, st.back(), -1); break;
                case Q_TEE_TMP: if (st.empty()) return -1; slots[in.a] = st.back(); break;
                case Q_PUSH_TMP: { auto it = slots.find(in.a); if (it == slots.end()) return -1; st.push_back(it->second); break; }
                default: return -1;
            }
        }
        return st.size() == 1 ? st[0] : -1;
    }

    bool has_product(int32_t v) {
        if ((size_t)v >= prod_memo.size()) prod_memo.resize(nd.size(), -1);
        if (prod_memo[v] >= 0) return prod_memo[v] != 0;
        const Node& n = nd[v];
        bool r = n.op == Q_MUL || n.op == Q_SQUARE || n.op == Q_MUL_CONST;
        if (n.op == OP_HSUM) { r = hs[n.a].size() > 1; for (const Item& it : hs[n.a]) r = r || has_product(it.node); }
        else {
            if (!r && n.x >= 0) r = has_product(n.x);
            if (!r && n.y >= 0) r = has_product(n.y);
        }
        if ((size_t)v >= prod_memo.size()) prod_memo.resize(nd.size(), -1);
        prod_memo[v] = r ? 1 : 0;
        return r;
    }

    uint32_t tree_cost(int32_t v) {
        if (cost_memo.size() < nd.size()) cost_memo.resize(nd.size(), 0);
        if (cost_memo[v]) return cost_memo[v] - 1;
        const Node& n = nd[v];
        uint64_t c = n.op == Q_MUL || n.op == Q_SQUARE || n.op == Q_MUL_CONST;
        if (n.op == OP_HSUM) { c = hs[n.a].size() - 1; for (const Item& it : hs[n.a]) c += tree_cost(it.node); }
        else { if (n.x >= 0) c += tree_cost(n.x); if (n.y >= 0) c += tree_cost(n.y); }
        cost_memo[v] = (uint32_t)std::min<uint64_t>(c, 0xFFFFFFF0u) + 1;
        return cost_memo[v] - 1;
    }
    // stack slots the evaluation of a node needs when the deeper operand of a commutative operation goes first (Sethi-Ullman):
    // operands are ORDERED by node number for hashing, which says nothing about the order they are best evaluated in -- a sum
    // p_0 + p_1 + ... + p_39 emitted leaf-first would hold forty values on the stack
    std::vector<uint16_t> need_memo;
    uint32_t need(int32_t v) {
        if (need_memo.size() < nd.size()) need_memo.resize(nd.size(), 0);
        if (need_memo[v]) return need_memo[v];
        const Node& n = nd[v];
        uint32_t r = 1;
        if (n.op == OP_HSUM) { for (const Item& it : hs[n.a]) r = std::max(r, need(it.node) + 1); }
        else if (n.y >= 0) {
            const uint32_t a = need(n.x), b = need(n.y);
            r = (n.op == Q_SUB) ? std::max(a, b + 1) : (a == b ? a + 1 : std::max(a, b));
        } else if (n.x >= 0) r = need(n.x);
        need_memo[v] = (uint16_t)std::min<uint32_t>(r, 0xFFFF);
        return need_memo[v];
    }
    // the Horner sum of `items` (sorted by constraint index on return), regrouped by common factors
    int32_t build_sum(std::vector<Item> items, int level, uint32_t* last_out) {
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.idx < b.idx; });
        if (level < 6 && items.size() >= 2) {
            std::unordered_map<int32_t, uint32_t> count;
            for (const Item& it : items) {
                const Node& n = nd[it.node];
                if (n.op != Q_MUL) continue;
                ++count[n.x];
                if (n.y != n.x) ++count[n.y];
            }
            // every product picks the factor most terms share; a group forms where at least two terms picked the same one
            std::vector<int32_t> pick(items.size(), -1);
            std::unordered_map<int32_t, uint32_t> picked;
            for (size_t i = 0; i < items.size(); ++i) {
                const Node& n = nd[items[i].node];
                if (n.op != Q_MUL) continue;
                const uint32_t cx = count[n.x], cy = count[n.y];
                if (std::max(cx, cy) < 2) continue;
                pick[i] = cy > cx ? n.y : n.x;
                ++picked[pick[i]];
            }
            bool any = false;
            for (size_t i = 0; i < items.size(); ++i) {
                if (pick[i] >= 0 && picked[pick[i]] < 2) pick[i] = -1;
                any = any || pick[i] >= 0;
            }
            if (any) {
                std::vector<Item> next;
                std::unordered_map<int32_t, std::vector<Item>> members;
                std::vector<int32_t> order;                      // factors in the order of their first term
                for (size_t i = 0; i < items.size(); ++i) {
                    if (pick[i] < 0) { next.push_back(items[i]); continue; }
                    const Node& n = nd[items[i].node];
                    const int32_t co = n.x == pick[i] ? n.y : n.x;
                    if (!members.count(pick[i])) order.push_back(pick[i]);
                    members[pick[i]].push_back({co, items[i].idx});
                }
                for (int32_t f : order) {
                    uint32_t last_g = 0;
                    const int32_t s = build_sum(members[f], level + 1, &last_g);
                    next.push_back({fresh(Q_MUL, 0, 0, f, s), last_g});
                    ++stats.groups;
                }
                items.swap(next);
                std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.idx < b.idx; });
            }
        }
        *last_out = items.back().idx;
        if (items.size() == 1) return items[0].node;
        hs.push_back(std::move(items));
        return fresh(OP_HSUM, (uint32_t)hs.size() - 1, 0, -1, -1);
    }

    // ---- emission
    std::vector<uint32_t> uses;
    std::vector<int32_t> vslot;          // node -> virtual slot once parked
    Prog out;
    // Chunked emission (round 6, for the sliced launches of csrc/quotient.hip: plan_slices): a parked value is only visible inside the top-level term that
    // parked it -- the next term recomputes (and, if it reads the value twice itself, parks) it again.  Every top-level FOLD is then a point where nothing
    // parked is alive, i.e. where the evaluator may cut the program into slices that run side by side.  `uses` are counted per term for the same reason.
    bool chunked = false;
    void count_uses_from(const std::vector<int32_t>& roots) {
        uses.assign(nd.size(), 0);
        std::vector<int32_t> work(roots);
        while (!work.empty()) {
            const int32_t v = work.back();
            work.pop_back();
            if (uses[v]++) continue;
            const Node& n = nd[v];
            if (n.op == OP_HSUM) for (const Item& it : hs[n.a]) work.push_back(it.node);
            else { if (n.x >= 0) work.push_back(n.x); if (n.y >= 0) work.push_back(n.y); }
        }
    }
    void count_uses(int32_t root) {
        uses.assign(nd.size(), 0);
        std::vector<int32_t> work{root};
        while (!work.empty()) {
            const int32_t v = work.back();
            work.pop_back();
            if (uses[v]++) continue;
            const Node& n = nd[v];
            if (n.op == OP_HSUM) for (const Item& it : hs[n.a]) work.push_back(it.node);
            else { if (n.x >= 0) work.push_back(n.x); if (n.y >= 0) work.push_back(n.y); }
        }
    }
    void emit(int32_t v, bool top) {
        if (vslot[v] >= 0) { out.push_back({Q_PUSH_TMP, (uint32_t)vslot[v], 0}); return; }
        const Node n = nd[v];
        switch (n.op) {
            case Q_PUSH_COL: out.push_back({Q_PUSH_COL, n.a, n.b}); return;
            case Q_PUSH_CONST: out.push_back({Q_PUSH_CONST, n.a, 0}); return;
            case Q_SUB: emit(n.x, false); emit(n.y, false); out.push_back({n.op, 0, 0}); break;
            case Q_ADD: case Q_MUL: {
                const bool swap_ = need(n.y) > need(n.x);
                emit(swap_ ? n.y : n.x, false); emit(swap_ ? n.x : n.y, false); out.push_back({n.op, 0, 0});
                break;
            }
            case Q_NEG: case Q_SQUARE: case Q_DOUBLE: emit(n.x, false); out.push_back({n.op, 0, 0}); break;
            case Q_MUL_CONST: case Q_ADD_CONST: emit(n.x, false); out.push_back({n.op, n.a, 0}); break;
            case OP_HSUM: {
                const std::vector<Item> items = hs[n.a];      // copy: emit() below may grow hs? (it does not, but keep references out of it)
                if (top) {
                    // the class's own sum goes through the accumulator: acc = acc * y^gap + item; items of one index are added first
                    size_t i = 0;
                    uint32_t prev = 0;
                    bool first = true;
                    while (i < items.size()) {
                        size_t j = i;
                        if (chunked) {
                            std::vector<int32_t> roots;
                            for (size_t q = i; q < items.size() && items[q].idx == items[i].idx; ++q) roots.push_back(items[q].node);
                            count_uses_from(roots);
                            std::fill(vslot.begin(), vslot.end(), -1);
                        }
                        while (j < items.size() && items[j].idx == items[i].idx) { emit(items[j].node, false); if (j > i) out.push_back({Q_ADD, 0, 0}); ++j; }
                        out.push_back({Q_FOLD, first ? C_ONE : C_YPOW0 + (items[i].idx - prev), 0});
                        prev = items[i].idx;
                        first = false;
                        i = j;
                    }
                    return;
                }
                for (size_t i = 0; i < items.size(); ++i) {
                    if (i && items[i].idx != items[i - 1].idx) out.push_back({Q_MUL_CONST, C_YPOW0 + (items[i].idx - items[i - 1].idx), 0});
                    emit(items[i].node, false);
                    if (i) out.push_back({Q_ADD, 0, 0});
                }
                break;
            }
            default: break;
        }
        if (uses[v] > 1 && has_product(v) && !no_park[v]) {
            vslot[v] = (int32_t)stats.parked++;
            out.push_back({Q_TEE_TMP, (uint32_t)vslot[v], 0});
        }
    }
    // virtual slots (one per parked node) -> physical slots by liveness
    void assign_slots() {
        std::vector<size_t> last_read(stats.parked, 0);
        for (size_t i = 0; i < out.size(); ++i) if (out[i].op == Q_PUSH_TMP) last_read[out[i].a] = i;
        std::vector<uint32_t> phys(stats.parked, 0), free_list;
        std::vector<std::vector<uint32_t>> dying(out.size() + 1);     // virtual slots whose last read is instruction i
        uint32_t next_phys = 0, live = 0;
        for (size_t i = 0; i < out.size(); ++i) {
            Instr& in = out[i];
            if (in.op == Q_TEE_TMP) {
                const uint32_t v = in.a;
                if (free_list.empty()) phys[v] = next_phys++;
                else { std::pop_heap(free_list.begin(), free_list.end(), std::greater<uint32_t>()); phys[v] = free_list.back(); free_list.pop_back(); }
                in.a = phys[v];
                ++live;
                stats.max_live = std::max(stats.max_live, live);
                if (last_read[v] <= i) { free_list.push_back(phys[v]); std::push_heap(free_list.begin(), free_list.end(), std::greater<uint32_t>()); --live; }     // never read (cannot happen: uses > 1)
            } else if (in.op == Q_PUSH_TMP) {
                const uint32_t v = in.a;
                in.a = phys[v];
                if (last_read[v] == i) { free_list.push_back(phys[v]); std::push_heap(free_list.begin(), free_list.end(), std::greater<uint32_t>()); --live; }
            }
        }
    }
};

// terms of one class (constraint index, program) -> its program; false: the caller keeps the old assembly
static bool compile_class(const std::vector<ClassTerm>& terms, uint32_t K, Prog& out, uint32_t* last, ClassCompileStats* stats_out = nullptr) {
    if (terms.empty() || K >= 0xFFFFu) return false;
    ClassCompiler cc;
    std::unordered_map<uint32_t, int32_t> slots;
    std::vector<ClassCompiler::Item> items;
    items.reserve(terms.size());
    for (const ClassTerm& t : terms) {
        const int32_t v = cc.read(t.prog, slots);
        if (v < 0 || t.cons >= K) return false;
        items.push_back({v, t.cons});
    }
    uint32_t last_idx = 0;
    int32_t root = cc.build_sum(std::move(items), 0, &last_idx);
    if (cc.nd[root].op != ClassCompiler::OP_HSUM) {          // a single item: wrap it so that the top-level emission closes with a FOLD
        cc.hs.push_back({{root, last_idx}});
        root = cc.fresh(ClassCompiler::OP_HSUM, (uint32_t)cc.hs.size() - 1, 0, -1, -1);
    }
    cc.count_uses(root);
    {   // large programs are emitted term by term so that the evaluator can slice them (ZK_QUOTIENT_CHUNK=0: one parking scope for the whole class)
        const char* e = getenv("ZK_QUOTIENT_CHUNK");
        cc.chunked = e ? atoi(e) != 0 : cc.nd.size() >= 4096;
    }
    cc.prod_memo.assign(cc.nd.size(), -1);
    cc.no_park.assign(cc.nd.size(), 0);
    // Parking pays by (uses - 1) x (products the value cost): tau = t + beta of a lookup table is ONE product and is read by every
    // lookup into the table -- parked; cell * 256 read again once, thousands of instructions later, would hold 32 B x rows of the
    // parking area all that time to save one product -- recomputed.  So: emit, measure every parked value's span, emit again without
    // the long-lived values whose benefit is below the price of admission; the price starts at one product and doubles until at most
    // CLASS_MAX_LIVE values are alive at once.
    constexpr uint32_t CLASS_MAX_LIVE = 64;
    uint64_t price = 1;
    for (uint32_t round = 0;; ++round) {
        cc.out.clear();
        cc.stats.parked = 0;
        cc.stats.max_live = 0;
        cc.vslot.assign(cc.nd.size(), -1);
        cc.emit(root, true);
        std::vector<int32_t> node_of(cc.stats.parked, -1);
        for (size_t v = 0; v < cc.vslot.size(); ++v) if (cc.vslot[v] >= 0) node_of[cc.vslot[v]] = (int32_t)v;
        std::vector<size_t> def_at(cc.stats.parked, 0), last_read(cc.stats.parked, 0);
        for (size_t i = 0; i < cc.out.size(); ++i) {
            if (cc.out[i].op == Q_TEE_TMP) def_at[cc.out[i].a] = i;
            else if (cc.out[i].op == Q_PUSH_TMP) last_read[cc.out[i].a] = i;
        }
        cc.assign_slots();
        if (round >= 24) break;
        if (round > 0 && cc.stats.max_live <= CLASS_MAX_LIVE) break;
        const size_t span_max = round < 20 ? 256 : 0;               // the last rounds evict whatever is still in the way
        bool changed = false;
        for (uint32_t sl = 0; sl < node_of.size(); ++sl) {
            if (node_of[sl] < 0 || last_read[sl] - def_at[sl] <= span_max) continue;
            const uint64_t benefit = (uint64_t)(cc.uses[node_of[sl]] - 1) * cc.tree_cost(node_of[sl]);
            if (benefit <= price) { cc.no_park[node_of[sl]] = 1; changed = true; }
        }
        if (!changed && cc.stats.max_live <= CLASS_MAX_LIVE) break;
        price *= 2;
    }
    int sp = 0, mx = 0;
    for (const Instr& in : cc.out) {
        if (in.op == Q_PUSH_COL || in.op == Q_PUSH_CONST || in.op == Q_PUSH_TMP) ++sp;
        else if (in.op == Q_ADD || in.op == Q_SUB || in.op == Q_MUL || in.op == Q_FOLD) --sp;
        mx = std::max(mx, sp);
    }
    if (sp != 0 || mx > 14) { if (getenv("ZK_QUOTIENT_TRACE")) fprintf(stderr, "[zk quotient] compile_class: stack depth %d (final %d): kept in the old form\n", mx, sp); return false; }
    for (const Instr& in : cc.out) if (in.op == Q_MUL || in.op == Q_SQUARE || in.op == Q_MUL_CONST || in.op == Q_FOLD) ++cc.stats.products;
    cc.stats.nodes = (uint32_t)cc.nd.size();
    cc.stats.instrs = (uint32_t)cc.out.size();
    cc.stats.depth = mx;
    out.swap(cc.out);
    *last = last_idx;
    if (stats_out) *stats_out = cc.stats;
    return true;
}
