/* numa.c — where the process runs on a host with more than one memory node: on the processors of the node its GPU hangs off. */
#define _GNU_SOURCE
#include "itx_host.h"

#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static cpu_set_t g_before;          /* the affinity the process was started with */
static int g_narrowed;

/* ITX_CPUS=<list like 64-127,192-255>: the processors this process (and every thread it starts) may run on — the way to keep
 * the reader's threads and the page-locked chunk buffers they first touch on the memory node the GPU hangs off */
static int cpus_from_env(void)
{
    const char *e = getenv("ITX_CPUS");
    if (!e || !*e) return 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    int any = 0;
    while (*e) {
        char *end;
        long a = strtol(e, &end, 10), b = a;
        if (end == e) break;
        if (*end == '-') {
            e = end + 1;
            b = strtol(e, &end, 10);
            if (end == e) break;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) {
                CPU_SET((int)c, &set);
                any = 1;
            }
        e = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') break;
    }
    if (any && sched_setaffinity(0, sizeof set, &set) != 0) {
        fprintf(stderr, "ITX_CPUS: cannot set the affinity, going on without\n");
        return 0;
    }
    return any;
}

/* The memory node the GPU this process will use hangs off, found without starting the runtime: the KFD topology lists the
 * GPUs in the runtime's order with their render minors; the ones whose render node this process may open are its devices.
 * -1: unknown (no such files, or a *_VISIBLE_DEVICES list that renumbers several devices). */
static int gpu_memory_node(int device)
{
    /* a *_VISIBLE_DEVICES list renumbers the devices: then only a process with ONE GPU to its name knows which one it means */
    const int renumbered = getenv("HIP_VISIBLE_DEVICES") || getenv("ROCR_VISIBLE_DEVICES") || getenv("CUDA_VISIBLE_DEVICES");
    long minors[64];
    int n = 0;
    for (int k = 0; k < 256 && n < 64; k++) {
        char path[128], line[256];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", k);
        FILE *f = fopen(path, "r");
        if (!f) {
            if (k < 2) continue;                                   /* (node numbers start at 0) */
            break;
        }
        long simd = 0, minor = -1;
        while (fgets(line, sizeof line, f)) {
            if (!strncmp(line, "simd_count ", 11)) simd = atol(line + 11);
            else if (!strncmp(line, "drm_render_minor ", 17)) minor = atol(line + 17);
        }
        fclose(f);
        if (simd <= 0 || minor < 0) continue;                      /* a CPU node, or another tenant's GPU (its properties do not read) */
        snprintf(path, sizeof path, "/dev/dri/renderD%ld", minor);
        const int fd = open(path, O_RDWR | O_CLOEXEC);            /* (open, not access: a device cgroup says no only here) */
        if (fd < 0) continue;                                      /* another tenant's */
        close(fd);
        minors[n++] = minor;
    }
    if (n == 0 || (renumbered && n != 1)) return -1;
    if (renumbered) device = 0;
    if (device < 0 || device >= n) return -1;
    char path[128];
    snprintf(path, sizeof path, "/sys/class/drm/renderD%ld/device/numa_node", minors[device]);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

/* Every thread of the process on the processors of that node (ITX_NUMA=0: wherever the scheduler likes, as before): the
 * rmsk parse, the table build, the reader's copies into the page-locked chunk buffers (first touched, hence placed, on
 * that node) and the bigWig writer stop migrating between the sockets. Measured on a two-socket box, 5 runs each of the
 * 500 M-read command: 2.36 s median on the GPU's node, 2.38 s on the other one, 2.52 s free to roam (and two of those five
 * above 2.9 s) — profiles/r03_cli_500M_hiseq_numa.json. */
static void stay_on_gpu_node(void)
{
    const char *e = getenv("ITX_NUMA");
    if ((e && atoi(e) == 0) || getenv("ITX_CPUS")) return;
    const int node = gpu_memory_node(getenv("ITX_DEVICE") ? atoi(getenv("ITX_DEVICE")) : getenv("ITX_RANK") ? atoi(getenv("ITX_RANK")) : 0);
    if (node < 0) return;
    char path[96], list[1024];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return;
    const int ok = fgets(list, sizeof list, f) != NULL;
    fclose(f);
    if (!ok) return;
    list[strcspn(list, "\n")] = 0;
    cpu_set_t now, want;
    if (sched_getaffinity(0, sizeof now, &now) != 0) return;
    CPU_ZERO(&want);
    int n = 0;
    for (const char *p = list; *p;) {
        char *end;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') {
            p = end + 1;
            b = strtol(p, &end, 10);
            if (end == p) break;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &now)) {
                CPU_SET((int)c, &want);
                n++;
            }
        if (*end != ',') break;
        p = end + 1;
    }
    if (n >= 4 && sched_setaffinity(0, sizeof want, &want) == 0) g_narrowed = 1;    /* (fewer than that allowed there: not worth it) */
}


static void report(const char *how)
{
    if (!getenv("ITX_NUMA_REPORT")) return;
    cpu_set_t now;
    if (sched_getaffinity(0, sizeof now, &now) != 0) return;
    int lo = -1, hi = -1;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &now)) {
            if (lo < 0) lo = c;
            hi = c;
        }
    fprintf(stderr, "[itx numa] %s: %d processors, %d .. %d\n", how, CPU_COUNT(&now), lo, hi);
}

void numa_place(void)
{
    if (sched_getaffinity(0, sizeof g_before, &g_before) != 0) return;
    if (cpus_from_env()) {                                         /* the user's word: the ranks of a job inherit it as it is */
        report("ITX_CPUS");
        return;
    }
    stay_on_gpu_node();
    report(g_narrowed ? "the GPU's memory node" : "as started");
}

/* Around posix_spawn of another rank: the child inherits the calling thread's affinity and narrows it for its own GPU, which
 * may hang off the other node — it has to start from what this process was started with. */
void numa_spawn_begin(void)
{
    if (g_narrowed) (void)sched_setaffinity(0, sizeof g_before, &g_before);
}

void numa_spawn_end(void)
{
    if (g_narrowed) {
        g_narrowed = 0;
        stay_on_gpu_node();
    }
}
