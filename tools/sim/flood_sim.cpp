// Experiment tooling (NOT product, NOT oracle): pixel-level simulator of the flood's round schemes on a dumped frame
// (tools/sim/dump_frame.py), to count rounds, walked pixels, tile steps and the per-round critical path of candidate
// commit rules before any of them is written in HIP.  The ordered semantics are those of filter.cpp:110-153 /
// line_detector.cpp:92-122; every scheme is checked against the sequential result.
//
//   g++ -O2 -std=c++17 -o /tmp/flood_sim tools/sim/flood_sim.cpp && /tmp/flood_sim /tmp/sim4k <scheme>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

static std::vector<float> dx, dy, thr;
static std::vector<uint8_t> dmask;
static std::vector<int32_t> sidx, sbin;
static float st[8], ct[8];
static int W, H, NS;

template <class T>
static std::vector<T> load(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) {
        perror(p.c_str());
        exit(1);
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (fread(v.data(), 1, n, f) != (size_t)n) exit(1);
    fclose(f);
    return v;
}

static inline float resp(int q, int b) { return std::fabs(fmaf(dx[q], st[b], dy[q] * ct[b])); }

struct Walk {
    std::vector<int> px;
    int tiles = 0;
    int par[4] = {0, 0, 0, 0};  // steps of the walk's critical path with 1, 2, 4, 8 wavefronts on its frontier (g_par)
    int levels = 0;             // tile levels of the footprint, breadth-first from the seed's tile
};
static bool g_par = false;

static std::vector<int> g_seen;  // epoch marks
static int g_epoch = 0;
static std::vector<int> g_tile_seen;

// footprint of seed k w.r.t. the committed pixels (their dmask is cleared)
static void footprint(int k, const std::vector<uint8_t>& dm, Walk& out) {
    out.px.clear();
    out.tiles = 0;
    const int b = sbin[k];
    const float t = thr[k];
    const int s = sidx[k];
    if (!(((dm[s] >> b) & 1) && resp(s, b) > t)) return;
    ++g_epoch;
    g_seen[s] = g_epoch;
    out.px.push_back(s);
    const int tw = (W + 7) / 8;
    for (size_t head = 0; head < out.px.size(); ++head) {
        const int p = out.px[head];
        const int ti = (p / W / 8) * tw + (p % W) / 8;
        if (g_tile_seen[ti] != g_epoch) {
            g_tile_seen[ti] = g_epoch;
            out.tiles++;
        }
        const int r = p / W, c = p % W;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                if (!dr && !dc) continue;
                const int rr = r + dr, cc = c + dc;
                if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                const int q = rr * W + cc;
                if (g_seen[q] == g_epoch) continue;
                if (((dm[q] >> b) & 1) && resp(q, b) > t) {
                    g_seen[q] = g_epoch;
                    out.px.push_back(q);
                }
            }
    }
    if (!g_par) return;
    // tile graph of the footprint, breadth-first from the seed's tile: a level's tiles are independent steps
    std::unordered_map<int, int> id;
    std::vector<std::vector<int>> adj;
    auto tile_of = [&](int p) { return (p / W / 8) * tw + (p % W) / 8; };
    for (int p : out.px) {
        const int ti = tile_of(p);
        if (id.emplace(ti, (int)adj.size()).second) adj.emplace_back();
    }
    for (int p : out.px) {
        const int a = id[tile_of(p)];
        const int r = p / W, c = p % W;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int rr = r + dr, cc = c + dc;
                if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                const int q = rr * W + cc;
                if (g_seen[q] != g_epoch) continue;
                const int bq = id[tile_of(q)];
                if (bq != a && std::find(adj[a].begin(), adj[a].end(), bq) == adj[a].end()) adj[a].push_back(bq);
            }
    }
    std::vector<int> lvl(adj.size(), -1), cur{id[tile_of(s)]}, nxt;
    lvl[cur[0]] = 0;
    for (int w = 0; w < 4; ++w) out.par[w] = 0;
    out.levels = 0;
    while (!cur.empty()) {
        out.levels++;
        for (int w = 0; w < 4; ++w) out.par[w] += ((int)cur.size() + (1 << w) - 1) >> w;
        nxt.clear();
        for (int a : cur)
            for (int bq : adj[a])
                if (lvl[bq] < 0) {
                    lvl[bq] = 1;
                    nxt.push_back(bq);
                }
        cur.swap(nxt);
    }
}

// sequential reference
static std::vector<int> sequential(std::vector<int>& sizes) {
    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    sizes.assign(NS, 0);
    Walk w;
    for (int k = 0; k < NS; ++k) {
        if (label[sidx[k]] >= 0) continue;
        footprint(k, dm, w);
        for (int p : w.px) {
            label[p] = k;
            dm[p] = 0;
        }
        sizes[k] = (int)w.px.size();
    }
    return label;
}

struct RoundStat {
    long walked_px = 0, steps = 0;
    int par[4] = {0, 0, 0, 0};
    long same_steps = 0;
    int width_hist[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int same_walks = 0, longest_changed = 0;
    int longest = 0, longest_commit = 0, n_walk = 0, n_commit = 0, n_dead = 0, n_active = 0, phases = 1, longest_sum = 0;
};

static void print_round(int r, const RoundStat& s) {
    if (g_par) printf("round %d: %d walks (%ld steps) have the footprint of the round before; longest walk among the others %d\n", r, s.same_walks, s.same_steps, s.longest_changed);
    if (g_par) {
        printf("round %d: walks of more than 190 tiles by tiles per level (1, 2, ... 9+):", r);
        for (int i = 1; i < 10; ++i) printf(" %d", s.width_hist[i]);
        printf("\n");
    }
    if (g_par) printf("round %d: longest walk on 1 / 2 / 4 / 8 wavefronts: %d / %d / %d / %d steps\n", r, s.par[0], s.par[1], s.par[2], s.par[3]);
    printf("round %d: active %d, walked %d seeds, %ld px in %ld steps, longest walk %d (committing: %d), sum of per-phase longest %d, phases %d; commit %d, dead %d\n", r,
           s.n_active, s.n_walk, s.walked_px, s.steps, s.longest, s.longest_commit, s.longest_sum, s.phases, s.n_commit, s.n_dead);
}

// ---- schemes 3, 4: pixel-level commits -------------------------------------------------------------------------------
// A seed commits, every round, the part G of its footprint that is connected to what it already owns (or to its seed
// pixel) through pixels that carry no lower stamp: no lower alive seed can reach those, so they belong to its final
// flood whatever happens elsewhere.  It stays active while its footprint still has contested pixels.  In later rounds it
// explores from what it owns.  scheme 3: every alive seed walks its whole footprint every round; scheme 4: seeds whose
// pixel lies in a part committed in this round (by a lower seed) are dead without walking (perfect deferral).
static int g_win_first_shift = 0, g_win_growth = 2;
static bool g_seed_level = false;
static int pixel_level(int scheme, const std::vector<int>& ref, long labelled) {
    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    const int INF = 0x7fffffff;
    std::vector<int> stamp((size_t)W * H, INF);
    std::vector<int> active(NS);
    for (int k = 0; k < NS; ++k) active[k] = k;
    std::vector<std::vector<int>> own(NS);  // committed pixels of a seed that is still active
    std::vector<int> cmark((size_t)W * H, 0);
    long tot_px = 0, tot_steps = 0, first_px = 0, first_steps = 0;
    int rounds = 0, crit = 0, crit_first = 0;
    std::vector<uint8_t> has_walked(NS, 0);
    std::vector<int> q;
    while (!active.empty()) {
        ++rounds;
        int longest_first = 0;
        RoundStat rs;
        rs.n_active = (int)active.size();
        std::vector<int> touched, next;
        std::vector<std::pair<int, std::vector<int>>> commits;  // applied at the end of the round (walks see the round-start state)
        // staged window: only seeds below it walk (any prefix of the seed order is a valid window)
        long window = NS;
        if (g_win_first_shift > 0) {
            window = std::max<long>(1024, NS >> g_win_first_shift);
            for (int r = 1; r < rounds && window < NS; ++r) window <<= g_win_growth;
        }
        for (int k : active) {  // ascending
            if (k >= window) {
                if (label[sidx[k]] >= 0 && own[k].empty()) rs.n_dead++;
                else next.push_back(k);
                continue;
            }
            const int b = sbin[k];
            const float t = thr[k];
            const int s = sidx[k];
            const bool has_own = !own[k].empty();
            if (!has_own) {
                if (label[s] >= 0 || cmark[s] == rounds) {  // taken by an earlier flood (this round: only scheme 4 knows before walking)
                    if (label[s] >= 0 || scheme == 4) {
                        rs.n_dead++;
                        continue;
                    }
                }
                if (!(((dm[s] >> b) & 1) && resp(s, b) > t)) {
                    rs.n_dead++;
                    continue;
                }
            }
            // explore from the owned pixels (or the seed) through uncommitted acceptable pixels
            ++g_epoch;
            q.clear();
            std::vector<int> fpx;  // uncommitted pixels reached
            if (has_own) {
                for (int p : own[k]) {
                    g_seen[p] = g_epoch;
                    q.push_back(p);
                }
            } else {
                g_seen[s] = g_epoch;
                q.push_back(s);
                fpx.push_back(s);
            }
            int tiles = 0;
            const int tw = (W + 7) / 8;
            for (size_t head = 0; head < q.size(); ++head) {
                const int p = q[head];
                const int r = p / W, c = p % W;
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        if (!dr && !dc) continue;
                        const int rr = r + dr, cc = c + dc;
                        if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                        const int x = rr * W + cc;
                        if (g_seen[x] == g_epoch) continue;
                        if (((dm[x] >> b) & 1) && resp(x, b) > t) {
                            g_seen[x] = g_epoch;
                            q.push_back(x);
                            fpx.push_back(x);
                        }
                    }
            }
            for (int p : fpx) {
                const int ti = (p / W / 8) * tw + (p % W) / 8;
                if (g_tile_seen[ti] != g_epoch) {
                    g_tile_seen[ti] = g_epoch;
                    tiles++;
                }
            }
            rs.n_walk++;
            rs.walked_px += (long)fpx.size();
            rs.steps += tiles;
            rs.longest = std::max(rs.longest, tiles);
            if (!has_walked[k]) {  // a first walk: the dependent chain through memory; later ones can replay the saved tile list
                has_walked[k] = 1;
                first_px += (long)fpx.size();
                first_steps += tiles;
                longest_first = std::max(longest_first, tiles);
            }
            // G: reachable from the sources through pixels of fpx without a lower stamp
            bool contested = false;
            std::vector<int> G;
            ++g_epoch;
            q.clear();
            if (has_own) {
                for (int p : own[k]) {
                    g_seen[p] = g_epoch;
                    q.push_back(p);
                }
            } else if (stamp[s] > k) {
                g_seen[s] = g_epoch;
                q.push_back(s);
                G.push_back(s);
            }
            for (int p : fpx)
                if (stamp[p] < k) contested = true;
            for (size_t head = 0; head < q.size(); ++head) {
                const int p = q[head];
                const int r = p / W, c = p % W;
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        if (!dr && !dc) continue;
                        const int rr = r + dr, cc = c + dc;
                        if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                        const int x = rr * W + cc;
                        if (g_seen[x] == g_epoch) continue;
                        if (((dm[x] >> b) & 1) && resp(x, b) > t && stamp[x] > k) {
                            g_seen[x] = g_epoch;
                            q.push_back(x);
                            G.push_back(x);
                        }
                    }
            }
            for (int p : fpx) {
                if (stamp[p] == INF) touched.push_back(p);
                stamp[p] = std::min(stamp[p], k);
            }
            if (g_seed_level && contested) G.clear();  // seed-level commits: all or nothing
            for (int p : G) cmark[p] = rounds;
            if (!G.empty()) rs.longest_commit = std::max(rs.longest_commit, tiles);
            if (contested && rounds == 1 && getenv("SIM_DIAG")) {
                // who is it that stays?  own pixel stamped by a lower seed (dominated) or not; bin of the dominator
                static long n_dom = 0, n_dom_samebin = 0, n_free = 0, px_dom = 0, px_free = 0, n_dom_ownG = 0, px_dom_same = 0;
                const int o = stamp[s];
                if (!has_own && o < k) {
                    n_dom++;
                    px_dom += (long)fpx.size();
                    if (sbin[o] == b) n_dom_samebin++, px_dom_same += (long)fpx.size();
                } else {
                    n_free++;
                    px_free += (long)fpx.size();
                }
                if (k == active.back() || (n_dom + n_free) % 2000 == 0)
                    fprintf(stderr, "diag: contested alive so far %ld dominated (%ld same bin as dominator; px %ld / %ld), %ld undominated (px %ld)\n", n_dom, n_dom_samebin, px_dom, px_dom_same, n_free, px_free);
            }
            if (contested) {
                next.push_back(k);
                if (!G.empty()) own[k].insert(own[k].end(), G.begin(), G.end());
            } else {
                rs.n_commit++;
                own[k].clear();
                own[k].shrink_to_fit();
            }
            commits.emplace_back(k, std::move(G));
        }
        for (auto& c : commits)
            for (int p : c.second) {
                label[p] = c.first;
                dm[p] = 0;
            }
        for (int p : touched) stamp[p] = INF;
        rs.longest_sum = rs.longest;
        print_round(rounds, rs);
        tot_px += rs.walked_px;
        tot_steps += rs.steps;
        crit += rs.longest;
        crit_first += longest_first;
        printf("   first walks of this round: longest %d\n", longest_first);
        active.swap(next);
        if (rounds > 200) break;
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != label[i];
    printf("scheme %d: %d rounds, %ld px walked (%.2fx labelled), %ld steps, critical path %d steps; FIRST walks only: %ld px (%.2fx), %ld steps, critical path %d; label mismatches vs sequential: %ld\n", scheme,
           rounds, tot_px, (double)tot_px / labelled, tot_steps, crit, first_px, (double)first_px / labelled, first_steps, crit_first, bad);
    return bad != 0;
}

// ---- scheme 9: ascending batches inside a round, pixel-level commits ------------------------------------------------------
// A round takes the active seeds in `nb` batches of ascending index (geometric sizes: the first holds n >> shift0 seeds,
// each next one `grow` times as many).  The seeds of a batch walk concurrently against the labels committed so far --
// including what the earlier batches of this round committed -- and stamp; then every seed of the batch commits the part
// G of its footprint that is connected to what it owns (or its seed pixel) through pixels without a lower stamp (stamps
// of the earlier batches' still-contested seeds stay in place).  A seed whose pixel has been committed when its batch
// starts is dead without a walk.  Contested seeds go on to the next round, where they explore from what they own.
// seed_level = 1: a seed commits all of its footprint or nothing (no G).
static int batched(int shift0, int grow, bool seed_level, const std::vector<int>& ref, long labelled) {
    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    const int INF = 0x7fffffff;
    std::vector<int> stamp((size_t)W * H, INF);
    std::vector<int> active(NS);
    for (int k = 0; k < NS; ++k) active[k] = k;
    std::vector<std::vector<int>> own(NS);
    long tot_px = 0, tot_steps = 0, g_px = 0;
    int rounds = 0, crit = 0, phases = 0;
    std::vector<int> q;
    const int tw = (W + 7) / 8;
    while (!active.empty() && rounds < 100) {
        ++rounds;
        std::vector<int> next, touched;
        long r_px = 0, r_steps = 0;
        int r_crit = 0, r_walk = 0, r_dead = 0, r_done = 0, r_batches = 0;
        size_t pos = 0;
        long bsize = std::max<long>(1024, (long)active.size() >> shift0);
        while (pos < active.size()) {
            const size_t end = std::min(active.size(), pos + (size_t)bsize);
            bsize *= grow;
            ++r_batches;
            int longest = 0;
            struct W1 {
                int k;
                std::vector<int> fpx;
            };
            std::vector<W1> walks;
            for (size_t ai = pos; ai < end; ++ai) {
                const int k = active[ai];
                const int b = sbin[k];
                const float t = thr[k];
                const int s = sidx[k];
                const bool has_own = !own[k].empty();
                if (!has_own) {
                    if (label[s] >= 0) {
                        r_dead++;
                        continue;
                    }
                    if (!(((dm[s] >> b) & 1) && resp(s, b) > t)) {
                        r_dead++;
                        continue;
                    }
                }
                ++g_epoch;
                q.clear();
                W1 w;
                w.k = k;
                if (has_own) {
                    for (int p : own[k]) {
                        g_seen[p] = g_epoch;
                        q.push_back(p);
                    }
                } else {
                    g_seen[s] = g_epoch;
                    q.push_back(s);
                    w.fpx.push_back(s);
                }
                for (size_t head = 0; head < q.size(); ++head) {
                    const int p = q[head];
                    const int r = p / W, c = p % W;
                    for (int dr = -1; dr <= 1; ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            if (!dr && !dc) continue;
                            const int rr = r + dr, cc = c + dc;
                            if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                            const int x = rr * W + cc;
                            if (g_seen[x] == g_epoch) continue;
                            if (((dm[x] >> b) & 1) && resp(x, b) > t) {
                                g_seen[x] = g_epoch;
                                q.push_back(x);
                                w.fpx.push_back(x);
                            }
                        }
                }
                int tiles = 0;
                for (int p : w.fpx) {
                    const int ti = (p / W / 8) * tw + (p % W) / 8;
                    if (g_tile_seen[ti] != g_epoch) {
                        g_tile_seen[ti] = g_epoch;
                        tiles++;
                    }
                }
                r_walk++;
                r_px += (long)w.fpx.size();
                r_steps += tiles;
                longest = std::max(longest, tiles);
                walks.push_back(std::move(w));
            }
            // stamps of the whole batch, then the commits of the batch
            for (auto& w : walks)
                for (int p : w.fpx) {
                    if (stamp[p] == INF) touched.push_back(p);
                    stamp[p] = std::min(stamp[p], w.k);
                }
            for (auto& w : walks) {
                const int k = w.k, b = sbin[k];
                const float t = thr[k];
                bool contested = false;
                for (int p : w.fpx)
                    if (stamp[p] < k) contested = true;
                std::vector<int> G;
                if (!contested) {
                    G = w.fpx;
                } else if (!seed_level) {
                    ++g_epoch;
                    q.clear();
                    if (!own[k].empty()) {
                        for (int p : own[k]) {
                            g_seen[p] = g_epoch;
                            q.push_back(p);
                        }
                    } else if (stamp[sidx[k]] >= k) {
                        g_seen[sidx[k]] = g_epoch;
                        q.push_back(sidx[k]);
                        G.push_back(sidx[k]);
                    }
                    for (size_t head = 0; head < q.size(); ++head) {
                        const int p = q[head];
                        const int r = p / W, c = p % W;
                        for (int dr = -1; dr <= 1; ++dr)
                            for (int dc = -1; dc <= 1; ++dc) {
                                if (!dr && !dc) continue;
                                const int rr = r + dr, cc = c + dc;
                                if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                                const int x = rr * W + cc;
                                if (g_seen[x] == g_epoch) continue;
                                if (((dm[x] >> b) & 1) && resp(x, b) > t && stamp[x] >= k) {
                                    g_seen[x] = g_epoch;
                                    q.push_back(x);
                                    G.push_back(x);
                                }
                            }
                    }
                    g_px += (long)G.size();
                }
                for (int p : G) {
                    label[p] = k;
                    dm[p] = 0;
                }
                if (contested) {
                    next.push_back(k);
                    own[k].insert(own[k].end(), G.begin(), G.end());
                } else {
                    r_done++;
                    own[k].clear();
                    own[k].shrink_to_fit();
                }
            }
            r_crit += longest;
            pos = end;
        }
        for (int p : touched) stamp[p] = INF;
        std::sort(next.begin(), next.end());
        printf("round %d: active %zu in %d batches, walked %d seeds, %ld px in %ld steps, sum of the batches' longest walks %d; finished %d, dead %d\n", rounds,
               active.size(), r_batches, r_walk, r_px, r_steps, r_crit, r_done, r_dead);
        tot_px += r_px;
        tot_steps += r_steps;
        crit += r_crit;
        phases += r_batches;
        active.swap(next);
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != label[i];
    printf("batched (first n>>%d, x%d, %s commits): %d rounds, %d batch launches, %ld px walked (%.2fx labelled) + %ld px of G re-walks (%.2fx), %ld steps, critical path %d steps; label mismatches vs sequential: %ld\n",
           shift0, grow, seed_level ? "seed-level" : "pixel-level", rounds, phases, tot_px, (double)tot_px / labelled, g_px, (double)g_px / labelled, tot_steps, crit, bad);
    return bad != 0;
}

// scheme 0: every active seed walks every round (what kernels_flood.hip did in r02)
// ---- scheme 11: NO rounds (dataflow), seed-level commits -------------------------------------------------------------------
// Every seed walks once at the start; from then on a seed walks again only when a commit took pixels out of its footprint
// (footprints only shrink, so an untouched footprint stays exact and its stamps stay valid as blockers).  A seed commits
// as soon as (i) its last walk is complete and untouched since, (ii) no lower active seed's stamps lie on its footprint,
// (iii) every lower active seed HAS stamps in place (its first walk is over): from that moment on nothing a lower seed does
// can reach the footprint, because those seeds' footprints only shrink.  While a seed walks again its old stamps (minus
// what has been committed) stay in place -- a superset of the new footprint, hence conservative.  Time is counted in tile
// steps; parallelism is unbounded (a lower bound of the critical path) or capped at `conc` walks in flight (strongest
// first, as the kernel's dispatcher would).  eager = 1: an invalidated seed walks again at once; 0: only when nothing lower
// blocks what is left of its old footprint (less work, staler blockers).
#include <queue>
static int g_pre_rounds = 0;  // bulk-synchronous rounds (scheme 0) in front of the dataflow phase
static int dataflow(int eager, long conc, const std::vector<int>& ref, long labelled) {
    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    std::vector<char> pre_gone(NS, 0);
    long pre_px = 0, pre_steps = 0, pre_crit = 0;
    for (int r = 0; r < g_pre_rounds; ++r) {  // every active seed walks, the unblocked ones commit, all stamps are erased
        const int INF = 0x7fffffff;
        std::vector<int> stamp((size_t)W * H, INF);
        std::vector<Walk> fp(NS);
        int longest = 0;
        for (int k = 0; k < NS; ++k) {
            if (pre_gone[k]) continue;
            if (label[sidx[k]] >= 0) {
                pre_gone[k] = 1;
                continue;
            }
            footprint(k, dm, fp[k]);
            pre_px += (long)fp[k].px.size();
            pre_steps += fp[k].tiles;
            longest = std::max(longest, fp[k].tiles);
            if (fp[k].px.empty()) pre_gone[k] = 1;
            for (int p : fp[k].px) stamp[p] = std::min(stamp[p], k);
        }
        pre_crit += longest;
        for (int k = 0; k < NS; ++k) {
            if (pre_gone[k]) continue;
            bool free_ = true;
            for (int p : fp[k].px)
                if (stamp[p] < k) {
                    free_ = false;
                    break;
                }
            if (!free_) continue;
            for (int p : fp[k].px) {
                label[p] = k;
                dm[p] = 0;
            }
            pre_gone[k] = 1;
        }
    }
    if (g_pre_rounds) printf("   %d round(s) first: %ld px, %ld steps, critical path %ld steps\n", g_pre_rounds, pre_px, pre_steps, pre_crit);
    struct Seed {
        std::vector<int> px;       // stamps in place (last complete footprint minus committed pixels)
        int tiles = 0;
        int nblock = 0;            // pixels of px whose lowest stamp belongs to a lower seed
        bool active = true, walking = false, invalid = false, has = false;
        long finish = 0;
    };
    std::vector<Seed> S(NS);
    // pixel -> seeds stamping it (ascending); only stamped pixels have an entry
    std::unordered_map<int, std::vector<int>> at;
    at.reserve((size_t)1 << 23);
    auto add_stamp = [&](int p, int k) {
        auto& v = at[p];
        v.insert(std::lower_bound(v.begin(), v.end(), k), k);
    };
    long tot_px = 0, tot_steps = 0, n_walks = 0, now = 0, last_commit = 0;
    int n_active = NS;
    struct Ev {
        long t;
        int k;
        bool operator<(const Ev& o) const { return t > o.t || (t == o.t && k > o.k); }
    };
    std::priority_queue<Ev> pq;
    std::vector<int> waiting;  // seeds that want to walk but found no free slot (capped parallelism): ascending
    long in_flight = 0;
    Walk w;
    // prefix gate (iii): the moment every seed below k has finished its first walk
    std::vector<long> first_finish(NS, -1);
    std::vector<int> ready;  // candidates to look at for a commit
    auto start_walk = [&](int k) {
        S[k].walking = true;
        S[k].invalid = false;
        footprint(k, dm, w);  // against what is committed NOW
        S[k].tiles = w.tiles;
        // the new footprint is kept aside until the walk is over
        S[k].finish = now + std::max(1, w.tiles);
        ++n_walks;
        tot_px += (long)w.px.size();
        tot_steps += w.tiles;
        in_flight++;
        pq.push({S[k].finish, k});
        return w.px;  // (copy)
    };
    std::vector<std::vector<int>> pending(NS);  // footprint of the walk in flight
    auto want_walk = [&](int k) {
        if (!S[k].active || S[k].walking) return;
        if (in_flight < conc) pending[k] = start_walk(k);
        else waiting.push_back(k);
    };
    auto remove_stamps = [&](int k, const std::vector<int>& px, std::vector<int>& unblocked) {
        for (int p : px) {
            auto it = at.find(p);
            if (it == at.end()) continue;
            auto& v = it->second;
            auto pos = std::lower_bound(v.begin(), v.end(), k);
            if (pos == v.end() || *pos != k) continue;
            const bool was_min = pos == v.begin();
            if (!was_min) S[k].nblock--;  // (k was blocked here; the pixel no longer counts for it)
            v.erase(pos);
            if (was_min && !v.empty()) {  // the new lowest stamp is no longer blocked HERE
                const int m = v.front();
                if (--S[m].nblock == 0) unblocked.push_back(m);
            }
            if (v.empty()) at.erase(it);
        }
    };
    std::vector<char> first_done(NS, 0);
    for (int k = 0; k < NS; ++k) {
        if (pre_gone[k] || label[sidx[k]] >= 0) {
            S[k].active = false;
            --n_active;
            first_done[k] = 1;
            continue;
        }
        want_walk(k);
    }
    std::sort(waiting.begin(), waiting.end());
    long gate_done = 0;  // seeds [0, gate_done) have finished their first walk
    while (gate_done < NS && first_done[gate_done]) ++gate_done;
    auto can_commit = [&](int k) {
        return S[k].active && S[k].has && !S[k].walking && !S[k].invalid && S[k].nblock == 0 && gate_done > k;
    };
    std::vector<int> unblocked;
    auto try_commit = [&](int k0) {
        std::vector<int> stack{k0};
        while (!stack.empty()) {
            const int k = stack.back();
            stack.pop_back();
            if (!can_commit(k)) continue;
            // commit: every pixel of the footprint is k's
            S[k].active = false;
            --n_active;
            last_commit = now;
            std::vector<int> px;
            px.swap(S[k].px);
            std::vector<int> hit;  // seeds that lose pixels
            for (int p : px) {
                label[p] = k;
                dm[p] = 0;
                auto it = at.find(p);
                if (it != at.end()) {
                    for (int j : it->second)
                        if (j != k) hit.push_back(j);
                }
            }
            unblocked.clear();
            remove_stamps(k, px, unblocked);
            std::sort(hit.begin(), hit.end());
            hit.erase(std::unique(hit.begin(), hit.end()), hit.end());
            for (int j : hit) {
                if (!S[j].active) continue;
                if (label[sidx[j]] >= 0) {  // its own pixel is taken: skipped for ever
                    S[j].active = false;
                    --n_active;
                    std::vector<int> pj;
                    pj.swap(S[j].px);
                    remove_stamps(j, pj, unblocked);
                    continue;
                }
                // drop the committed pixels from its stamps (they are k's now); what is left stays as a conservative blocker
                std::vector<int> keep, gone;
                for (int p : S[j].px) (label[p] >= 0 ? gone : keep).push_back(p);
                remove_stamps(j, gone, unblocked);
                S[j].px.swap(keep);
                S[j].invalid = true;
                if (eager || S[j].nblock == 0) want_walk(j);
            }
            for (int m : unblocked) {
                if (S[m].active && S[m].invalid && !S[m].walking) want_walk(m);  // (lazy policy: now is the time)
                stack.push_back(m);
            }
        }
    };
    while (!pq.empty()) {
        const Ev e = pq.top();
        pq.pop();
        now = e.t;
        const int k = e.k;
        in_flight--;
        S[k].walking = false;
        if (S[k].active) {
            // the walk is over: its footprint replaces the old stamps -- unless a commit took pixels out of it meanwhile
            std::vector<int> np;
            np.swap(pending[k]);
            bool touched = false;
            for (int p : np)
                if (label[p] >= 0) touched = true;
            unblocked.clear();
            std::vector<int> old;
            old.swap(S[k].px);
            remove_stamps(k, old, unblocked);
            if (np.empty()) {  // the seed pixel is no longer acceptable: nothing to claim
                S[k].active = false;
                --n_active;
            } else {
                std::vector<int> keep;
                for (int p : np)
                    if (label[p] < 0) keep.push_back(p);
                S[k].px.swap(keep);
                S[k].nblock = 0;
                for (int p : S[k].px) {
                    add_stamp(p, k);
                    const auto& v = at[p];
                    if (v.front() < k) S[k].nblock++;
                    else if (v.size() > 1 && v[1] > k) {  // k became the lowest here: the former lowest is blocked here now
                        S[v[1]].nblock++;
                    }
                }
                S[k].has = true;
                S[k].invalid = touched;
                if (label[sidx[k]] >= 0) {
                    S[k].active = false;
                    --n_active;
                    std::vector<int> pj;
                    pj.swap(S[k].px);
                    remove_stamps(k, pj, unblocked);
                } else if (touched && (eager || S[k].nblock == 0)) {
                    want_walk(k);
                }
            }
            if (!first_done[k]) {
                first_done[k] = 1;
                while (gate_done < NS && first_done[gate_done]) ++gate_done;
            }
            std::vector<int> ub = unblocked;
            try_commit(k);
            for (int m : ub) {
                if (S[m].active && S[m].invalid && !S[m].walking) want_walk(m);
                try_commit(m);
            }
            // the gate may have opened for seeds that were only waiting for it
            static long gate_seen = 0;
            if (gate_done > gate_seen) {
                for (long j = gate_seen; j < gate_done; ++j) try_commit((int)j);
                gate_seen = gate_done;
            }
        } else if (!first_done[k]) {
            first_done[k] = 1;
            while (gate_done < NS && first_done[gate_done]) ++gate_done;
        }
        // free slots go to the strongest waiting seeds
        if (!waiting.empty() && in_flight < conc) {
            std::sort(waiting.begin(), waiting.end());
            waiting.erase(std::unique(waiting.begin(), waiting.end()), waiting.end());
            size_t i = 0;
            for (; i < waiting.size() && in_flight < conc; ++i) {
                const int j = waiting[i];
                if (S[j].active && !S[j].walking) pending[j] = start_walk(j);
            }
            waiting.erase(waiting.begin(), waiting.begin() + (long)i);
        }
        if (pq.empty() && n_active > 0) {  // nobody walks and somebody is left: look once more at everyone
            for (int j = 0; j < NS; ++j)
                if (S[j].active) {
                    if (S[j].invalid && !S[j].walking) want_walk(j);
                    try_commit(j);
                }
        }
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != label[i];
    printf("scheme 11 (dataflow, %s re-walks, %s): %ld walks, %ld px walked (%.2fx labelled), %ld steps; last commit at t = %ld steps, last walk over at t = %ld; seeds left active %d; label mismatches vs sequential: %ld\n",
           eager ? "eager" : "lazy", conc >= (long)NS ? "unbounded parallelism" : ("at most " + std::to_string(conc) + " walks in flight").c_str(), n_walks, tot_px,
           (double)tot_px / labelled, tot_steps, last_commit, now, n_active, bad);
    return bad != 0 || n_active != 0;
}


// ---- scheme 12: rounds, but a seed whose last footprint lost no pixel to a commit does not walk again -------------------------
// A footprint changes only through commits of its own pixels (acceptance is static, what borders a footprint is unacceptable
// or committed for good).  So a blocked seed whose logged footprint is untouched re-stamps the log -- parallel atomics, no
// dependent chain -- and its round is EXACTLY what a walk would have given.  `min_tiles`: only walks of at least so many
// tiles keep a log.  Reported per round: the longest walk that was needed against the longest walk of the plain schedule.
static int rounds_lazy(int min_tiles, const std::vector<int>& ref, long labelled) {
    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    std::vector<char> gone(NS, 0);
    std::vector<Walk> fp(NS);
    std::vector<char> has(NS, 0);
    const int INF = 0x7fffffff;
    long steps_all = 0, steps_needed = 0, crit_all = 0, crit_needed = 0, restamp_tiles = 0;
    int rounds = 0, n_active = NS;
    while (n_active > 0) {
        std::vector<int> stamp((size_t)W * H, INF);
        int longest_all = 0, longest_needed = 0, skipped = 0, walked = 0, longest_restamp = 0;
        for (int k = 0; k < NS; ++k) {
            if (gone[k]) continue;
            if (label[sidx[k]] >= 0) {
                gone[k] = 1;
                --n_active;
                continue;
            }
            bool reuse = false;
            if (has[k]) {
                reuse = true;
                for (int p : fp[k].px)
                    if (label[p] >= 0) {
                        reuse = false;
                        break;
                    }
            }
            if (!reuse) {
                footprint(k, dm, fp[k]);
                has[k] = fp[k].tiles >= min_tiles;
                steps_needed += fp[k].tiles;
                longest_needed = std::max(longest_needed, fp[k].tiles);
                ++walked;
            } else {
                ++skipped;
                restamp_tiles += fp[k].tiles;
                longest_restamp = std::max(longest_restamp, fp[k].tiles);
            }
            steps_all += fp[k].tiles;
            longest_all = std::max(longest_all, fp[k].tiles);
            if (fp[k].px.empty()) {
                gone[k] = 1;
                --n_active;
            }
            for (int p : fp[k].px) stamp[p] = std::min(stamp[p], k);
        }
        int committed = 0;
        for (int k = 0; k < NS; ++k) {
            if (gone[k]) continue;
            bool free_ = true;
            for (int p : fp[k].px)
                if (stamp[p] < k) {
                    free_ = false;
                    break;
                }
            if (!free_) continue;
            for (int p : fp[k].px) {
                label[p] = k;
                dm[p] = 0;
            }
            gone[k] = 1;
            --n_active;
            ++committed;
        }
        ++rounds;
        crit_all += longest_all;
        crit_needed += longest_needed;
        printf("  round %d: walked %d (longest %d), re-stamped from the log %d (longest log %d tiles); plain schedule's longest walk %d; committed %d, left %d\n",
               rounds, walked, longest_needed, skipped, longest_restamp, longest_all, committed, n_active);
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != label[i];
    printf("scheme 12 (rounds, untouched footprints of >= %d tiles are re-stamped, not walked): %d rounds; steps %ld -> %ld; sum of the rounds' longest walks %ld -> %ld; %ld tiles re-stamped; label mismatches vs sequential: %ld\n",
           min_tiles, rounds, steps_all, steps_needed, crit_all, crit_needed, restamp_tiles, bad);
    return bad != 0;
}


// ---- scheme 13: rounds with the LAZY rule of scheme 11 ---------------------------------------------------------------------------
// A blocked seed remembers the lowest seed whose stamp lay on its footprint.  While that seed is unresolved it does not walk:
// it re-stamps its logged footprint minus what has been committed since (a superset of its present footprint, hence a
// conservative blocker for higher seeds; itself it stays blocked).  When the blocker is resolved it walks again.  A seed that
// comes out unblocked on a log that commits have touched walks in the next round before it may commit.
// `min_tiles`: only walks of at least so many tiles are treated this way (the others walk every round).
static int rounds_parked(int min_tiles, const std::vector<int>& ref, long labelled) {
    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    std::vector<char> gone(NS, 0), has(NS, 0), stale(NS, 0);
    std::vector<int> blocker(NS, -1);
    std::vector<Walk> fp(NS);
    const int INF = 0x7fffffff;
    long steps_all = 0, steps_needed = 0, crit_needed = 0, restamp_tiles = 0;
    int rounds = 0, n_active = NS;
    while (n_active > 0 && rounds < 200) {
        std::vector<int> stamp((size_t)W * H, INF);
        int longest_needed = 0, skipped = 0, walked = 0, longest_restamp = 0;
        for (int k = 0; k < NS; ++k) {
            if (gone[k]) continue;
            if (label[sidx[k]] >= 0) {
                gone[k] = 1;
                --n_active;
                continue;
            }
            const bool park = has[k] && blocker[k] >= 0 && !gone[blocker[k]];
            if (!park) {
                footprint(k, dm, fp[k]);
                has[k] = fp[k].tiles >= min_tiles;
                stale[k] = 0;
                steps_needed += fp[k].tiles;
                longest_needed = std::max(longest_needed, fp[k].tiles);
                ++walked;
            } else {
                std::vector<int> keep;
                for (int p : fp[k].px)
                    if (label[p] < 0) keep.push_back(p);
                if (keep.size() != fp[k].px.size()) stale[k] = 1;
                fp[k].px.swap(keep);
                ++skipped;
                restamp_tiles += fp[k].tiles;
                longest_restamp = std::max(longest_restamp, fp[k].tiles);
            }
            if (fp[k].px.empty()) {
                gone[k] = 1;
                --n_active;
            }
            for (int p : fp[k].px) stamp[p] = std::min(stamp[p], k);
        }
        int committed = 0;
        for (int k = 0; k < NS; ++k) {
            if (gone[k]) continue;
            int low = INF;
            for (int p : fp[k].px) low = std::min(low, stamp[p]);
            if (low < k) {
                blocker[k] = low;
                continue;
            }
            blocker[k] = -1;
            if (stale[k]) continue;  // unblocked on a stale log: walks next round
            for (int p : fp[k].px) {
                label[p] = k;
                dm[p] = 0;
            }
            gone[k] = 1;
            --n_active;
            ++committed;
        }
        ++rounds;
        crit_needed += longest_needed;
        printf("  round %d: walked %d (longest %d), re-stamped %d (longest log %d tiles); committed %d, left %d\n", rounds, walked, longest_needed, skipped,
               longest_restamp, committed, n_active);
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != label[i];
    printf("scheme 13 (rounds, lazy rule for walks of >= %d tiles): %d rounds; steps %ld; sum of the rounds' longest walks %ld; %ld tiles re-stamped; left %d; label mismatches vs sequential: %ld\n",
           min_tiles, rounds, steps_needed, crit_needed, restamp_tiles, n_active, bad);
    (void)steps_all;
    return bad != 0;
}

// scheme 1: perfect deferral -- seeds that die in the round never walk or stamp (lower bound of the deferral family)
// scheme 2: opportunistic deferral in dispatch batches of `conc` walks (strongest first), phases until nothing is left
//           to walk; a deferred seed is one whose own pixel carries a lower stamp when its batch starts
int main(int argc, char** argv) {
    const std::string dir = argv[1];
    const int scheme = argc > 2 ? atoi(argv[2]) : 0;
    g_par = std::getenv("SIM_PAR") != nullptr;
    const int conc = argc > 3 ? atoi(argv[3]) : 5120;
    const int max_phases = argc > 4 ? atoi(argv[4]) : 100;
    FILE* m = fopen((dir + "/meta.txt").c_str(), "r");
    if (fscanf(m, "%d %d %d", &W, &H, &NS) != 3) return 1;
    fclose(m);
    dx = load<float>(dir + "/dx.f32");
    dy = load<float>(dir + "/dy.f32");
    dmask = load<uint8_t>(dir + "/dmask.u8");
    sidx = load<int32_t>(dir + "/seed_idx.i32");
    sbin = load<int32_t>(dir + "/seed_bin.i32");
    thr = load<float>(dir + "/seed_thr.f32");
    auto tr = load<float>(dir + "/trig.f32");
    for (int i = 0; i < 8; ++i) st[i] = tr[i], ct[i] = tr[8 + i];
    g_seen.assign((size_t)W * H, 0);
    g_tile_seen.assign((size_t)((W + 7) / 8) * ((H + 7) / 8), 0);

    std::vector<int> ref_sizes;
    const std::vector<int> ref = sequential(ref_sizes);
    long labelled = 0;
    for (int v : ref) labelled += v >= 0;
    printf("%dx%d, %d seeds, %ld labelled px\n", W, H, NS, labelled);
    if (scheme == 12) return rounds_lazy(argc > 3 ? atoi(argv[3]) : 0, ref, labelled);
    if (scheme == 13) return rounds_parked(argc > 3 ? atoi(argv[3]) : 0, ref, labelled);
    if (scheme == 11) {
        if (argc > 5) g_pre_rounds = atoi(argv[5]);
        return dataflow(argc > 3 ? atoi(argv[3]) : 1, argc > 4 ? atol(argv[4]) : (long)NS, ref, labelled);
    }
    if (scheme == 9 || scheme == 10) return batched(argc > 3 ? atoi(argv[3]) : 3, argc > 4 ? atoi(argv[4]) : 2, scheme == 10, ref, labelled);
    if (scheme >= 3 && scheme <= 6) {
        // 3: pixel-level, all walk; 4: + perfect deferral; 5: seed-level commits + perfect deferral (= scheme 1); 6: seed-level, all walk
        if (argc > 3) g_win_first_shift = atoi(argv[3]);
        if (argc > 4) g_win_growth = atoi(argv[4]);
        g_seed_level = scheme >= 5;
        return pixel_level(scheme == 5 ? 4 : (scheme == 6 ? 3 : scheme), ref, labelled);
    }

    std::vector<uint8_t> dm = dmask;
    std::vector<int> label((size_t)W * H, -1);
    std::vector<int> active(NS);
    for (int k = 0; k < NS; ++k) active[k] = k;
    const int INF = 0x7fffffff;
    std::vector<int> stamp((size_t)W * H, INF);
    std::vector<Walk> fp(NS);
    long tot_px = 0, tot_steps = 0;
    int crit = 0, crit_commit = 0, rounds = 0, tot_phases = 0;
    while (!active.empty()) {
        ++rounds;
        RoundStat rs;
        rs.n_active = (int)active.size();
        std::vector<int> commit, dead;
        std::vector<uint8_t> walked(NS, 0);
        std::vector<int> touched;  // pixels stamped this round
        auto do_walk = [&](int k) {
            const size_t prev_size = fp[k].px.size();
            footprint(k, dm, fp[k]);
            // (a footprint only shrinks from round to round: equal size = the same pixels = a walk a replay of the saved list could replace)
            if (rounds > 1 && fp[k].px.size() == prev_size && prev_size > 0) {
                rs.same_walks++;
                rs.same_steps += fp[k].tiles;
            } else {
                rs.longest_changed = std::max(rs.longest_changed, fp[k].tiles);
            }
            walked[k] = 1;
            rs.n_walk++;
            rs.walked_px += (long)fp[k].px.size();
            rs.steps += fp[k].tiles;
            rs.longest = std::max(rs.longest, fp[k].tiles);
            for (int w = 0; w < 4; ++w) rs.par[w] = std::max(rs.par[w], fp[k].par[w]);
            if (g_par && fp[k].tiles > 190) {  // tiles per level of the walks beyond the first storage tier
                const int r = std::min(9, fp[k].tiles / std::max(1, fp[k].levels));
                rs.width_hist[r]++;
            }
            for (int p : fp[k].px) {
                if (stamp[p] == INF) touched.push_back(p);
                stamp[p] = std::min(stamp[p], k);
            }
        };
        if (scheme == 7 || scheme == 8) {
            // seed-level commits as scheme 0, but only the seeds below a window walk.  7: static hold-back (argv[3] = pct:
            // the weakest (100 - pct) % wait until at most 64 seeds below the line are active); 8: dynamic -- after a round
            // the window closes at the lowest BLOCKED seed whose walk took more than argv[3] steps, while seeds below it
            // are active
            static long window = -1;
            static int phase = 0;
            const int param = argc > 3 ? atoi(argv[3]) : (scheme == 7 ? 80 : 96);
            if (window < 0) window = scheme == 7 ? (long)NS * param / 100 : NS;
            for (int k : active)
                if (k < window) do_walk(k);
            rs.longest_sum = rs.longest;
            int lowest_long_blocked = INF;
            for (int k : active) {
                if (k >= window) continue;
                if (fp[k].px.empty()) {
                    dead.push_back(k);
                    continue;
                }
                bool blocked = false;
                for (int p : fp[k].px)
                    if (stamp[p] < k) {
                        blocked = true;
                        break;
                    }
                if (!blocked) commit.push_back(k);
                else if (fp[k].tiles > param && scheme == 8) lowest_long_blocked = std::min(lowest_long_blocked, k);
            }
            // next window
            std::vector<uint8_t> gone2(NS, 0);
            for (int k : commit) gone2[k] = 1;
            for (int k : dead) gone2[k] = 1;
            if (scheme == 7) {
                long below = 0;
                for (int k : active)
                    if (k < window && !gone2[k]) below++;
                if (phase == 0 && below <= 64) {
                    window = NS;
                    phase = 1;
                }
            } else {
                long below = 0;
                for (int k : active)
                    if (k < lowest_long_blocked && !gone2[k]) below++;
                window = (lowest_long_blocked != INF && below > 0) ? lowest_long_blocked : NS;
                if (commit.empty() && dead.empty()) window = NS;
            }
        } else if (scheme == 0) {
            for (int k : active) do_walk(k);
            rs.longest_sum = rs.longest;
            for (int k : active) {
                if (fp[k].px.empty()) {
                    dead.push_back(k);  // accepts nothing, not even itself
                    continue;
                }
                bool blocked = false;
                for (int p : fp[k].px)
                    if (stamp[p] < k) {
                        blocked = true;
                        break;
                    }
                if (!blocked) commit.push_back(k);
            }
        } else if (scheme == 1) {
            // ascending order: a seed whose pixel lies in a footprint committed in this round is dead and leaves no stamp
            std::vector<uint8_t> ccov;  // marks via stamp2
            std::vector<int> cover((size_t)0);
            static std::vector<int> cmark;
            if (cmark.empty()) cmark.assign((size_t)W * H, 0);
            for (int k : active) {  // (active is ascending)
                if (cmark[sidx[k]] == rounds) {
                    dead.push_back(k);
                    continue;
                }
                footprint(k, dm, fp[k]);
                walked[k] = 1;
                rs.n_walk++;
                rs.walked_px += (long)fp[k].px.size();
                rs.steps += fp[k].tiles;
                rs.longest = std::max(rs.longest, fp[k].tiles);
                if (fp[k].px.empty()) {
                    dead.push_back(k);
                    continue;
                }
                bool blocked = false;
                for (int p : fp[k].px)
                    if (stamp[p] < k) {
                        blocked = true;
                        break;
                    }
                for (int p : fp[k].px) {
                    if (stamp[p] == INF) touched.push_back(p);
                    stamp[p] = std::min(stamp[p], k);
                }
                if (!blocked) {
                    commit.push_back(k);
                    for (int p : fp[k].px) cmark[p] = rounds;
                }
            }
            rs.longest_sum = rs.longest;
        } else {
            // opportunistic deferral.  status: 0 not yet considered, 1 walked, 2 deferred
            std::vector<uint8_t> status(NS, 0);
            std::vector<int> to_walk = active;
            rs.phases = 0;
            rs.longest_sum = 0;
            std::vector<uint8_t> blocked(NS, 0);
            std::vector<int> deferred;
            while (!to_walk.empty() && rs.phases < max_phases) {
                rs.phases++;
                int phase_longest = 0;
                // batches of `conc`: a seed sees the stamps of earlier batches (and earlier phases) only
                for (size_t b0 = 0; b0 < to_walk.size(); b0 += conc) {
                    const size_t b1 = std::min(to_walk.size(), b0 + (size_t)conc);
                    std::vector<int> go;
                    for (size_t i = b0; i < b1; ++i) {
                        const int k = to_walk[i];
                        if (stamp[sidx[k]] < k && rs.phases == 1) {  // its own pixel is reached by a lower seed: deferred
                            status[k] = 2;
                            deferred.push_back(k);
                        } else {
                            go.push_back(k);
                        }
                    }
                    for (int k : go) {
                        footprint(k, dm, fp[k]);
                        status[k] = 1;
                        walked[k] = 1;
                        rs.n_walk++;
                        rs.walked_px += (long)fp[k].px.size();
                        rs.steps += fp[k].tiles;
                        phase_longest = std::max(phase_longest, fp[k].tiles);
                    }
                    for (int k : go)
                        for (int p : fp[k].px) {
                            if (stamp[p] == INF) touched.push_back(p);
                            stamp[p] = std::min(stamp[p], k);
                        }
                }
                rs.longest = std::max(rs.longest, phase_longest);
                rs.longest_sum += phase_longest;
                // decide: tentative commits = walked, unblocked; deferred seeds whose pixel's lowest stamper commits are dead;
                // the other deferred seeds must walk in the next phase
                std::vector<uint8_t> tent(NS, 0);
                for (int k : active)
                    if (status[k] == 1 && !fp[k].px.empty()) {
                        bool bl = false;
                        for (int p : fp[k].px)
                            if (stamp[p] < k) {
                                bl = true;
                                break;
                            }
                        tent[k] = !bl;
                    }
                to_walk.clear();
                std::vector<int> still;
                for (int k : deferred) {
                    const int o = stamp[sidx[k]];
                    if (o < k && tent[o]) {
                        still.push_back(k);  // dead if o really commits: stays deferred
                    } else {
                        to_walk.push_back(k);
                    }
                }
                deferred.swap(still);
            }
            // final decision (barrier: the lowest active seed that neither walked nor is dead)
            std::vector<uint8_t> tent(NS, 0);
            for (int k : active)
                if (status[k] == 1 && !fp[k].px.empty()) {
                    bool bl = false;
                    for (int p : fp[k].px)
                        if (stamp[p] < k) {
                            bl = true;
                            break;
                        }
                    tent[k] = !bl;
                }
            int barrier = INF;
            for (int k : to_walk) barrier = std::min(barrier, k);  // phases exhausted: these never walked
            // deferred whose owner does not commit below the barrier are alive and un-walked: lower the barrier (fixpoint)
            for (bool again = true; again;) {
                again = false;
                for (int k : deferred) {
                    const int o = stamp[sidx[k]];
                    const bool owner_commits = o < k && tent[o] && o < barrier;
                    if (!owner_commits && k < barrier) {
                        barrier = k;
                        again = true;
                    }
                }
            }
            for (int k : active) {
                if (status[k] == 1 && fp[k].px.empty()) dead.push_back(k);
                else if (tent[k] && k < barrier) commit.push_back(k);
            }
        }
        for (int k : commit) {
            rs.longest_commit = std::max(rs.longest_commit, fp[k].tiles);
            for (int p : fp[k].px) {
                label[p] = k;
                dm[p] = 0;
            }
        }
        for (int p : touched) stamp[p] = INF;
        std::vector<int> next;
        std::vector<uint8_t> gone(NS, 0);
        for (int k : commit) gone[k] = 1;
        for (int k : dead) gone[k] = 1;
        for (int k : active) {
            if (gone[k]) continue;
            if (label[sidx[k]] >= 0) {
                rs.n_dead++;
                continue;
            }
            next.push_back(k);
        }
        rs.n_commit = (int)commit.size();
        rs.n_dead += (int)dead.size();
        print_round(rounds, rs);
        tot_px += rs.walked_px;
        tot_steps += rs.steps;
        crit += rs.longest_sum;
        crit_commit += rs.longest_commit;
        tot_phases += rs.phases;
        if (commit.empty() && next.size() == active.size() && scheme != 7 && scheme != 8) {
            printf("STALL\n");
            break;
        }
        if (rounds > 100) break;
        active.swap(next);
    }
    long bad = 0;
    for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != label[i];
    printf("scheme %d: %d rounds (%d phases), %ld px walked (%.2fx labelled), %ld steps, critical path %d steps (committing walks only: %d); label mismatches vs sequential: %ld\n",
           scheme, rounds, tot_phases, tot_px, (double)tot_px / labelled, tot_steps, crit, crit_commit, bad);
    return bad != 0;
}
