// sanitize_driver.cpp -- CPU-only harness for tools/sanitize_host.sh: the host grid builder (csrc/grid_host.cpp) and the
// native table packing (csrc/pack_host.cpp) compiled INTO this program with a sanitizer, run on a jittered hexahedron
// mesh and a Kuhn-tetrahedra mesh with 1 thread and with a full team, every array compared between the two runs
// (the builder replaces the reference's documented benign races, grid.pyx:438-444, 515-517, by relaxed atomics, per-row
// sorts and parallel sorts: the outputs must not depend on the schedule).  No GPU, no HIP: test infrastructure.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../ninpol_amd/csrc/grid_host.cpp"
#include "../ninpol_amd/csrc/pack_host.cpp"

using namespace nin;

static const int HEX_FACES[6][4] = {{0, 3, 2, 1}, {4, 5, 6, 7}, {0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}};
static const int TET_FACES[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
static const int HEX_EDGES[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
static const int TET_EDGES[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};

static void tables(HostGrid &h) {   // element types as in utils/point_ordering.yaml: 4 = tetra, 5 = hexahedron
    memset(h.npoel, 0, sizeof h.npoel); memset(h.nfael, 0, sizeof h.nfael); memset(h.nedel, 0, sizeof h.nedel);
    memset(h.lnofa, 0, sizeof h.lnofa); memset(h.lpofa, -1, sizeof h.lpofa); memset(h.lpoed, -1, sizeof h.lpoed);
    h.npoel[4] = 4; h.nfael[4] = 4; h.nedel[4] = 6;
    for (int f = 0; f < 4; ++f) { h.lnofa[4][f] = 3; for (int k = 0; k < 3; ++k) h.lpofa[4][f][k] = TET_FACES[f][k]; }
    for (int e = 0; e < 6; ++e) { h.lpoed[4][e][0] = TET_EDGES[e][0]; h.lpoed[4][e][1] = TET_EDGES[e][1]; }
    h.npoel[5] = 8; h.nfael[5] = 6; h.nedel[5] = 12;
    for (int f = 0; f < 6; ++f) { h.lnofa[5][f] = 4; for (int k = 0; k < 4; ++k) h.lpofa[5][f][k] = HEX_FACES[f][k]; }
    for (int e = 0; e < 12; ++e) { h.lpoed[5][e][0] = HEX_EDGES[e][0]; h.lpoed[5][e][1] = HEX_EDGES[e][1]; }
}

struct Built { HostGrid h; };

static int build(bool tets, int n, int threads, HostGrid &h) {
    const int s = n + 1;
    std::vector<double> xyz((size_t)s * s * s * 3);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-0.15 / n, 0.15 / n);
    for (int k = 0; k < s; ++k) for (int j = 0; j < s; ++j) for (int i = 0; i < s; ++i) {
        const size_t p = ((size_t)k * s + j) * s + i;
        xyz[3 * p] = (double)i / n + U(rng); xyz[3 * p + 1] = (double)j / n + U(rng); xyz[3 * p + 2] = (double)k / n + U(rng);
    }
    static const int KUHN[6][4] = {{0, 1, 2, 6}, {0, 2, 3, 6}, {0, 3, 7, 6}, {0, 7, 4, 6}, {0, 4, 5, 6}, {0, 5, 1, 6}};
    std::vector<int64_t> conn, types;
    for (int k = 0; k < n; ++k) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const int64_t n0 = ((int64_t)k * s + j) * s + i;
        const int64_t c[8] = {n0, n0 + 1, n0 + 1 + s, n0 + s, n0 + (int64_t)s * s, n0 + (int64_t)s * s + 1, n0 + (int64_t)s * s + 1 + s, n0 + (int64_t)s * s + s};
        if (!tets) { for (int q = 0; q < 8; ++q) conn.push_back(c[q]); types.push_back(5); }
        else for (auto &t : KUHN) { for (int q = 0; q < 4; ++q) conn.push_back(c[t[q]]); for (int q = 4; q < 8; ++q) conn.push_back(-1); types.push_back(4); }
    }
    h.dim = 3; h.n_elems = (int64_t)types.size(); h.n_points = (int64_t)s * s * s; h.build_edges = 1; h.num_threads = threads;
    tables(h);
    int rc = h.build(conn.data(), types.data(), xyz.data(), 3);
    if (rc) return rc;
    { const ScopedTeam team(threads); h.build_psup(); h.build_inedel(); }
    return 0;
}

template <class T> static bool same(const char *name, const std::vector<T> &a, const std::vector<T> &b) {
    if (a.size() == b.size() && (a.empty() || !memcmp(a.data(), b.data(), a.size() * sizeof(T)))) return true;
    fprintf(stderr, "MISMATCH between 1 thread and the full team: %s\n", name);
    return false;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20, team = argc > 2 ? atoi(argv[2]) : 8;
    int bad = 0;
    for (int tets = 0; tets < 2; ++tets) {
        HostGrid a, b;
        if (build(tets, n, 1, a) || build(tets, n, team, b)) { fprintf(stderr, "build failed\n"); return 2; }
#define CMP(f) bad += !same(#f, a.f, b.f)
        CMP(inpoel); CMP(etype); CMP(esup_ptr); CMP(esup); CMP(fsup_ptr); CMP(fsup); CMP(esuf_ptr); CMP(esuf); CMP(esuel); CMP(infael);
        CMP(inpofa); CMP(boundary_faces); CMP(boundary_points); CMP(coords); CMP(centroids); CMP(faces_centers); CMP(normal_faces);
        CMP(faces_areas); CMP(psup_ptr); CMP(psup); CMP(inedel); CMP(inpoed);
#undef CMP
        // table packing + hash on the same data
        std::vector<double> K((size_t)a.n_elems * 9), dm1((size_t)a.n_elems), dm2((size_t)a.n_elems), row((size_t)a.n_elems * 9);
        for (size_t i = 0; i < K.size(); ++i) K[i] = 1.0 + (double)(i % 9 % 4 == 0) + 1e-3 * (double)(i % 17);
        { const ScopedTeam t1(1); nin_diff_mag(K.data(), a.n_elems, dm1.data()); }
        { const ScopedTeam t2(team); nin_diff_mag(K.data(), a.n_elems, dm2.data()); nin_pack_table_row(K.data(), a.n_elems, 9, 9, row.data()); }
        bad += !same("diff_mag", dm1, dm2);
        bad += !same("table row", K, row);
        uint64_t h1 = 0, h2 = 0;
        { const ScopedTeam t1(1); nin_hash64(K.data(), K.size() * 8, &h1); }
        { const ScopedTeam t2(team); nin_hash64(K.data(), K.size() * 8, &h2); }
        if (h1 != h2) { fprintf(stderr, "MISMATCH: nin_hash64\n"); ++bad; }
        std::vector<uint8_t> f1((size_t)a.n_points), f2((size_t)a.n_points);
        { const ScopedTeam t1(1); pack_node_flags(a.coords.data(), a.boundary_points.data(), a.n_points, f1.data()); }
        { const ScopedTeam t2(team); pack_node_flags(a.coords.data(), a.boundary_points.data(), a.n_points, f2.data()); }
        bad += !same("node flags", f1, f2);
        printf("%s %d^3: E=%lld P=%lld F=%lld edges=%lld: 1 thread == %d threads on %d arrays%s\n", tets ? "Kuhn tets" : "hexahedra", n,
               (long long)a.n_elems, (long long)a.n_points, (long long)a.n_faces, (long long)a.n_edges, team, 26, bad ? " -- NO" : "");
    }
    return bad ? 1 : 0;
}
