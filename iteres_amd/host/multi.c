/* multi.c — one process per GPU (see itx_host.h). The reference is one thread with one set of counters
 * (generic.c:705-722); what makes N copies of its loop one job is that every output is a commutative integer sum
 * (SURVEY.md §8e): each rank scans its share of the input against its own replica of the table, one sum-reduce of the
 * compact partials onto rank 0 (RCCL over xGMI) ends the stream, rank 0 writes the files. */
#define _GNU_SOURCE
#include "itx_host.h"

#include <errno.h>
#include <pthread.h>
#include <signal.h>
#include <spawn.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

extern char **environ;

static int g_rank, g_world = 1, g_device, g_mode = ITX_COMM_RCCL, g_external;
static char g_id[512], g_prefix[32];
static int g_argc;
static char **g_argv;
#define MAX_RANKS 64
static pid_t kids[MAX_RANKS];
static int n_kids, kids_done;
static pthread_t watcher;
static int watcher_on;

int multi_rank(void) { return g_rank; }
int multi_world(void) { return g_world; }
int multi_device(void) { return g_device; }
int multi_comm_mode(void) { return g_mode; }
int multi_selftest(void) { return g_external && g_world == 1; }      /* ITX_COMM_SELFTEST: one rank, whole exchange */
const char *multi_comm_id(void) { return g_id; }

static void quiet_rank(void)
{
    /* banners and progress are rank 0's; this rank's errors and warnings still reach the terminal */
    const int efd = dup(2);
    if (efd >= 0) {
        itx_err_stream = fdopen(efd, "w");
        if (itx_err_stream) setvbuf(itx_err_stream, NULL, _IONBF, 0);
    }
    snprintf(g_prefix, sizeof g_prefix, "[rank %d] ", g_rank);
    itx_err_prefix = g_prefix;
    if (!freopen("/dev/null", "w", stderr)) { /* keep going with what we have */ }
}

void multi_early(int argc, char **argv)
{
    g_argc = argc;
    g_argv = argv;
    const char *r = getenv("ITX_RANK"), *w = getenv("ITX_WORLD");
    /* ITX_COMM_SELFTEST with ITX_WORLD=1: a job of one rank that still walks the whole exchange (communicator made beside the
     * scan, both reduces, finish from the reduced partial) — all of the N > 1 path a one-GPU box can run through RCCL */
    if (!r || !w || atoi(w) < 1 || (atoi(w) == 1 && !getenv("ITX_COMM_SELFTEST"))) return;
    g_world = atoi(w);
    g_rank = atoi(r);
    if (g_rank < 0 || g_rank >= g_world || g_world > MAX_RANKS) die("ITX_RANK=%s / ITX_WORLD=%s: not a rank of a job", r, w);
    g_device = getenv("ITX_DEVICE") ? atoi(getenv("ITX_DEVICE")) : g_rank;
    const char *id = getenv("ITX_COMM_ID");
    if (!id || !*id || strlen(id) >= sizeof g_id) die("ITX_RANK / ITX_WORLD are set but ITX_COMM_ID (a file path the ranks share) is not");
    strcpy(g_id, id);
    const char *x = getenv("ITX_EXCHANGE");
    g_mode = x && strcmp(x, "file") == 0 ? ITX_COMM_FILE : ITX_COMM_RCCL;
    g_external = 1;
    if (g_rank > 0) quiet_rank();
}

static void kill_kids(void)
{
    for (int i = 0; i < n_kids; i++)
        if (kids[i] > 0) kill(kids[i], SIGTERM);
}

/* rank 0: a rank that leaves with an error ends the job (the others would wait for it in the exchange). Only the ranks'
 * own pids are waited for: a waitpid(-1) here also reaped the decompressor children of tables.c, whose owner then could not
 * learn how they ended. */
static void *watch_main(void *arg)
{
    (void)arg;
    for (int left = n_kids; left > 0;) {
        int progressed = 0;
        for (int i = 0; i < n_kids; i++) {
            if (kids[i] <= 0) continue;
            int st = 0;
            const pid_t p = waitpid(kids[i], &st, WNOHANG);
            if (p == 0) continue;
            if (p < 0) {
                if (errno == EINTR) continue;
                kids[i] = 0;                                         /* not ours to wait for any more */
                left--;
                progressed = 1;
                continue;
            }
            kids[i] = 0;
            left--;
            progressed = 1;
            if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
                fprintf(itx_err_stream ? itx_err_stream : stderr, "[iteres] rank %d left with %s %d: the job ends here\n", i + 1,
                        WIFEXITED(st) ? "status" : "signal", WIFEXITED(st) ? WEXITSTATUS(st) : WTERMSIG(st));
                kill_kids();
                fflush(NULL);
                _exit(255);
            }
        }
        if (!progressed && left > 0) usleep(2000);
    }
    kids_done = 1;
    return NULL;
}

/* ITX_GPUS=all: the number of visible devices, counted by a short-lived child so that this process has not touched the
 * GPU when it starts the other ranks */
static int probe_devices(void)
{
    int fd[2];
    if (pipe(fd) != 0) return 1;
    const pid_t p = fork();
    if (p < 0) return 1;
    if (p == 0) {
        close(fd[0]);
        int n = itx_device_count();
        if (write(fd[1], &n, sizeof n) != (ssize_t)sizeof n) _exit(1);
        _exit(0);
    }
    close(fd[1]);
    int n = 1, st;
    if (read(fd[0], &n, sizeof n) != (ssize_t)sizeof n) n = 1;
    close(fd[0]);
    while (waitpid(p, &st, 0) < 0 && errno == EINTR) {}
    return n > 0 ? n : 1;
}

/* The GPUs this process may use, without touching one: the KFD topology lists every GPU of the host, but only those this
 * container was given can be read; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow it further. 0: cannot tell. */
static int count_gpus_sysfs(void)
{
    int n = 0;
    for (int node = 0; node < 256; node++) {
        char path[128];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
        FILE *f = fopen(path, "r");
        if (!f) {
            if (errno == ENOENT) break;
            continue;                                                /* another tenant's GPU */
        }
        char key[64];
        unsigned long long v;
        while (fscanf(f, "%63s %llu", key, &v) == 2)
            if (strcmp(key, "simd_count") == 0 && v > 0) n++;
        fclose(f);
    }
    const char *vis[2] = {getenv("HIP_VISIBLE_DEVICES"), getenv("ROCR_VISIBLE_DEVICES")};
    for (int k = 0; k < 2; k++)
        if (vis[k] && *vis[k]) {
            int m = 1;
            for (const char *c = vis[k]; *c; c++)
                if (*c == ',') m++;
            if (n == 0 || m < n) n = m;
        }
    return n;
}

size_t multi_min_share(void)
{
    const char *e = getenv("ITX_SPLIT_MIN");                        /* tests: tiny files are shared all the same */
    return e && atol(e) >= 1 ? (size_t)atol(e) : (size_t)64 << 20;
}

void multi_begin(int splittable, const char *aln_arg, int multi_file)
{
    if (g_external || g_world > 1) return;                          /* a launcher made the ranks */
    if (!splittable || !aln_arg) return;
    /* ITX_GPUS=N: that many ranks; unset or "all": every GPU this process may use — as far as the input is worth sharing
     * (every rank parses the tables and builds its replica: a share below ITX_SPLIT_MIN bytes, 64 MiB, is not worth a rank) */
    const char *g = getenv("ITX_GPUS");
    int n;
    if (g && *g && strcmp(g, "all") != 0) {
        n = atoi(g);
        /* no more ranks than devices (every extra rank would die in hipSetDevice and end the job) — unless a device map says
         * where the ranks go (ITX_GPU_MAP: rehearsals with ranks that share a card) */
        if (n > 1 && !(getenv("ITX_GPU_MAP") && *getenv("ITX_GPU_MAP"))) {
            int have = count_gpus_sysfs();
            if (have == 0) have = probe_devices();
            if (n > have) {
                warnf("[iteres] note: ITX_GPUS=%d but %d device%s visible: running on %d", n, have, have == 1 ? " is" : "s are", have);
                n = have;
            }
        }
    } else {
        n = count_gpus_sysfs();
        if (n == 0) n = probe_devices();
    }
    if (n > MAX_RANKS) n = MAX_RANKS;
    if (n <= 1) return;
    {
        char *copy = xstrdup(aln_arg), *save = NULL;
        size_t total = 0;
        int ok = 1;
        for (char *tok = multi_file ? strtok_r(copy, ",", &save) : copy; tok; tok = multi_file ? strtok_r(NULL, ",", &save) : NULL) {
            struct stat sb;
            if (stat(tok, &sb) != 0 || !S_ISREG(sb.st_mode)) {
                ok = 0;                                              /* a pipe, or a missing file: one rank reports it the reference's way */
                break;
            }
            total += (size_t)sb.st_size;
        }
        free(copy);
        if (!ok) return;
        const size_t worth = total / multi_min_share();
        if ((size_t)n > worth) n = (int)worth;
        if (n <= 1) return;
    }
    /* which device each rank takes: its own number, or ITX_GPU_MAP=a,b,c,...; ranks that share a device (a rehearsal on a
     * one-GPU box) exchange their partials through files, RCCL wants a GPU per rank */
    int map[MAX_RANKS], dup = 0;
    for (int r = 0; r < n; r++) map[r] = r;
    const char *m = getenv("ITX_GPU_MAP");
    if (m && *m) {
        char *copy = xstrdup(m), *save = NULL;
        int r = 0;
        for (char *tok = strtok_r(copy, ",", &save); tok && r < n; tok = strtok_r(NULL, ",", &save)) map[r++] = atoi(tok);
        free(copy);
    }
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (map[a] == map[b]) dup = 1;
    const char *x = getenv("ITX_EXCHANGE");
    g_mode = (dup || (x && strcmp(x, "file") == 0)) ? ITX_COMM_FILE : ITX_COMM_RCCL;
    const char *tmp = getenv("TMPDIR");
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(g_id, sizeof g_id, "%s/itx_comm_%ld_%ld%09ld.id", tmp && *tmp ? tmp : "/tmp", (long)getpid(), (long)ts.tv_sec, ts.tv_nsec);
    g_world = n;
    g_rank = 0;
    g_device = map[0];
    /* the other ranks: this program again, same arguments, told who they are */
    size_t ne = 0;
    while (environ[ne]) ne++;
    for (int r = 1; r < n; r++) {
        char **env = xcalloc(ne + 8, sizeof(char *));
        size_t k = 0;
        for (size_t i = 0; i < ne; i++)
            if (strncmp(environ[i], "ITX_GPUS=", 9) != 0) env[k++] = environ[i];
        char v[5][600];
        snprintf(v[0], sizeof v[0], "ITX_RANK=%d", r);
        snprintf(v[1], sizeof v[1], "ITX_WORLD=%d", n);
        snprintf(v[2], sizeof v[2], "ITX_DEVICE=%d", map[r]);
        snprintf(v[3], sizeof v[3], "ITX_COMM_ID=%s", g_id);
        snprintf(v[4], sizeof v[4], "ITX_EXCHANGE=%s", g_mode == ITX_COMM_FILE ? "file" : "rccl");
        for (int j = 0; j < 5; j++) env[k++] = v[j];
        env[k] = NULL;
        pid_t pid = 0;
        numa_spawn_begin();
        const int rc = posix_spawn(&pid, "/proc/self/exe", NULL, NULL, g_argv, env);
        numa_spawn_end();
        free(env);
        if (rc != 0) {
            kill_kids();
            die("cannot start rank %d: %s", r, strerror(rc));
        }
        kids[n_kids++] = pid;
    }
    itx_die_hook = kill_kids;
    if (pthread_create(&watcher, NULL, watch_main, NULL) == 0) watcher_on = 1;
}

void multi_finish(void)
{
    if (watcher_on) {
        pthread_join(watcher, NULL);
        watcher_on = 0;
    }
    if (g_world > 1 && g_rank == 0 && !g_external) unlink(g_id);
}
