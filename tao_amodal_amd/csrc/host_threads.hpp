// How many OpenMP threads the host-side readers / writers start.
//
// omp_get_max_threads() is the number of logical CPUs the process may run on;
// inside a container the CPU-time quota of the control group is usually far
// below that (the MI355X boxes of this project: 256 logical CPUs, cpu.max =
// 16 CPUs).  256 threads on a 16-CPU quota are throttled for most of every
// scheduler period: a byte scan of a 3.7 GB prediction file took 0.30-0.40 s a
// pass with 256 threads and 0.13 s with 32.  The count used is therefore
// min(OpenMP's maximum, CPUs in the affinity mask, ceil(quota / period));
// TAOAMD_HOST_THREADS overrides it.
#pragma once
#include <omp.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>

namespace taoamd {

inline int quota_cpus()
{
    // cgroup v2: "<quota> <period>" or "max <period>"
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long period = 0;
        const int got = fscanf(f, "%31s %lld", q, &period);
        fclose(f);
        if (got == 2 && period > 0 && q[0] != 'm') {
            const long long quota = atoll(q);
            if (quota > 0) return (int)((quota + period - 1) / period);
        }
        return 0;
    }
    // cgroup v1
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(f, "%lld", &quota) != 1) quota = -1;
        fclose(f);
    }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(f, "%lld", &period) != 1) period = 0;
        fclose(f);
    }
    if (quota > 0 && period > 0) return (int)((quota + period - 1) / period);
    return 0;
}

inline int host_threads()
{
    static const int n = [] {
        if (const char *e = getenv("TAOAMD_HOST_THREADS")) {
            const int v = atoi(e);
            if (v > 0) return v;
        }
        int t = omp_get_max_threads();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0 && a < t) t = a;
        }
        const int q = quota_cpus();
        if (q > 0 && q < t) t = q;
        return t < 1 ? 1 : t;
    }();
    return n;
}

// A caller's cap on the teams THIS host thread starts (taoamd_host_thread_cap;
// 0 = none): when several entry points run side by side -- the two readers of
// the CLI, beside an interpreter thread that imports torch -- more runnable
// threads than the control group's quota are all frozen together for the rest
// of every scheduler period, the serial thread on the critical path included.
inline thread_local int g_thread_cap = 0;

inline int team_threads()
{
    const int n = host_threads();
    return g_thread_cap > 0 && g_thread_cap < n ? g_thread_cap : n;
}

// the calling thread's OpenMP team size for the lifetime of the object
struct ThreadScope {
    int before;
    ThreadScope() : before(omp_get_max_threads())
    {
        omp_set_num_threads(team_threads());
    }
    ~ThreadScope() { omp_set_num_threads(before); }
    ThreadScope(const ThreadScope &) = delete;
    ThreadScope &operator=(const ThreadScope &) = delete;
};

}  // namespace taoamd
