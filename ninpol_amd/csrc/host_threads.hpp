// host_threads.hpp -- how many OpenMP threads the host-side loops should use by default (internal).
//
// omp_get_max_threads() counts the machine's logical CPUs.  In a container with a CPU QUOTA (cgroup cpu.max /
// cfs_quota_us: the GPU boxes give a job 16 of 256 logical CPUs that way) a team of that size is throttled for whole
// scheduler periods -- a 5 ms loop takes 200 ms.  The default team is therefore min(affinity mask, quota, OMP limit).
#pragma once
#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <cstdio>

namespace nin {

inline int default_host_threads() {
    int n = omp_get_max_threads();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                      // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') sscanf(q, "%lld", &quota);
        fclose(f);
    } else {                                                                     // cgroup v1
        if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = -1; fclose(fp); }
    }
    if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    return std::max(1, n);
}

// The team of the library's OWN parallel regions.  It is passed as num_threads(...) on every region and to the parallel
// sorts; the process-wide OpenMP default (omp_set_num_threads) is never touched -- the host application's other libgomp
// users (torch, scipy, the caller's code) keep theirs.  ScopedTeam: an explicit num_threads of nin_grid_create, for the
// duration of that build on the calling thread.
inline int &host_team_override() {
    static thread_local int t = 0;
    return t;
}
inline int host_team() {
    const int t = host_team_override();
    if (t > 0) return t;
    static const int d = default_host_threads();
    return d;
}
struct ScopedTeam {
    int prev;
    explicit ScopedTeam(int n) : prev(host_team_override()) { if (n > 0) host_team_override() = n; }
    ~ScopedTeam() { host_team_override() = prev; }
    ScopedTeam(const ScopedTeam &) = delete;
    ScopedTeam &operator=(const ScopedTeam &) = delete;
};

}  // namespace nin
