/* Design analysis (CPU): lock-step group walk with deferral of sparse descents.
 * A group of gs key-adjacent bodies walks the octree in lock-step (descend when any member opens).
 * When 1..T members open a node, the descent is not taken by the group: one task (body, subtree) per
 * opening member is queued and later walked by a single lane on its own.  Counts what each part costs.
 * Build: gcc -O2 -fopenmp -shared -fPIC -o walk_sim.so walk_sim.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 mask_t;

static int64_t solo_walk(int32_t root, const double *p, const double *half, const double *com,
                         const int32_t *children, const uint8_t *is_leaf, double theta, double eps2) {
    /* visits of one body below `root` (root itself already visited and opened) */
    int32_t st[512];
    int sp = 0;
    int64_t v = 0;
    for (int c = 0; c < 8; c++) { int32_t ch = children[8 * (int64_t)root + c]; if (ch >= 0) st[sp++] = ch; }
    while (sp > 0) {
        int32_t node = st[--sp];
        v++;
        if (is_leaf[node]) continue;
        double dx = com[3 * node] - p[0], dy = com[3 * node + 1] - p[1], dz = com[3 * node + 2] - p[2];
        double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
        if (!(half[node] * 2.0 / dist < theta))
            for (int c = 0; c < 8; c++) { int32_t ch = children[8 * (int64_t)node + c]; if (ch >= 0) st[sp++] = ch; }
    }
    return v;
}

static int cmp_desc(const void *a, const void *b) { int64_t x = *(const int64_t *)a, y = *(const int64_t *)b; return x < y ? 1 : (x > y ? -1 : 0); }

/* out: [0] lock-step visits, [1] lane-visits inside lock-step, [2] deferral events, [3] tasks,
 * [4] lane-visits in tasks, [5] drain wave-iterations (list scheduling on gs lanes, flush when >= flushq
 * tasks are queued, tasks taken in queue order), [6] number of drains, [7] max task length */
void walk_sim(const double *pos, const int64_t *order, int64_t n, int gs, int T, int flushq, const double *half,
              const double *com, const int32_t *children, const uint8_t *is_leaf, double theta, double softening,
              int64_t *out) {
    const double eps2 = softening * softening;
    int64_t ngroups = n / gs;
    int64_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0, o5 = 0, o6 = 0, o7 = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : o0, o1, o2, o3, o4, o5, o6) reduction(max : o7)
    for (int64_t g = 0; g < ngroups; g++) {
        int64_t lo = g * gs;
        int cap = 8192, sp = 0;
        int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * cap);
        mask_t *sm = (mask_t *)malloc(sizeof(mask_t) * cap);
        int qcap = 1 << 16, qn = 0;
        int64_t *q = (int64_t *)malloc(sizeof(int64_t) * qcap);
        int64_t laneload[128];
        sn[0] = 0; sm[0] = (gs == 128) ? ~(mask_t)0 : ((((mask_t)1) << gs) - 1); sp = 1;
        for (;;) {
            if (sp == 0 || qn >= flushq) {
                if (qn > 0) {
                    /* drain: greedy pull in queue order onto gs lanes */
                    memset(laneload, 0, sizeof(laneload));
                    for (int t = 0; t < qn; t++) {
                        int best = 0;
                        for (int l = 1; l < gs; l++) if (laneload[l] < laneload[best]) best = l;
                        laneload[best] += q[t];
                    }
                    int64_t mk = 0;
                    for (int l = 0; l < gs; l++) if (laneload[l] > mk) mk = laneload[l];
                    o5 += mk; o6 += 1; qn = 0;
                }
                if (sp == 0) break;
            }
            sp--;
            int32_t node = sn[sp];
            mask_t mask = sm[sp];
            o0++;
            mask_t open = 0;
            int a = 0, o = 0;
            for (int l = 0; l < gs; l++) {
                if (!((mask >> l) & 1)) continue;
                a++;
                if (is_leaf[node]) continue;
                int64_t i = order[lo + l];
                double dx = com[3 * node] - pos[3 * i], dy = com[3 * node + 1] - pos[3 * i + 1], dz = com[3 * node + 2] - pos[3 * i + 2];
                double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                if (!(half[node] * 2.0 / dist < theta)) { open |= ((mask_t)1) << l; o++; }
            }
            o1 += a;
            if (o == 0) continue;
            if (o <= T) {
                o2++;
                for (int l = 0; l < gs; l++) {
                    if (!((open >> l) & 1)) continue;
                    int64_t i = order[lo + l];
                    int64_t v = solo_walk(node, pos + 3 * i, half, com, children, is_leaf, theta, eps2);
                    if (qn == qcap) { qcap *= 2; q = (int64_t *)realloc(q, sizeof(int64_t) * qcap); }
                    q[qn++] = v;
                    o3++; o4 += v;
                    if (v > o7) o7 = v;
                }
                continue;
            }
            for (int c = 0; c < 8; c++) {
                int32_t ch = children[8 * (int64_t)node + c];
                if (ch >= 0) {
                    if (sp == cap) { cap *= 2; sn = (int32_t *)realloc(sn, sizeof(int32_t) * cap); sm = (mask_t *)realloc(sm, sizeof(mask_t) * cap); }
                    sn[sp] = ch; sm[sp] = open; sp++;
                }
            }
        }
        free(sn); free(sm); free(q);
    }
    (void)cmp_desc;
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3; out[4] = o4; out[5] = o5; out[6] = o6; out[7] = o7;
}

/* Composition of a lock-step group walk: out[0] visits on cells, [1] visits on leaves, [2] visits on leaves whose
 * parent has only leaves as children ("twig"), [3] lane-visits on cells, [4] on leaves, [5] on twig leaves,
 * [6] twig openings (group visits of a twig that some member opens), [7] cells that are twigs among visited cells */
void walk_mix(const double *pos, const int64_t *order, int64_t n, int gs, const double *half, const double *com,
              const int32_t *children, const uint8_t *is_leaf, double theta, double softening, int64_t *out) {
    const double eps2 = softening * softening;
    int64_t ngroups = n / gs;
    int64_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma omp parallel
    {
        int64_t q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 8)
        for (int64_t g = 0; g < ngroups; g++) {
            int64_t lo = g * gs;
            int cap = 8192, sp = 0;
            int32_t *sn = (int32_t *)malloc(sizeof(int32_t) * cap);
            mask_t *sm = (mask_t *)malloc(sizeof(mask_t) * cap);
            uint8_t *st = (uint8_t *)malloc(cap);  /* parent was a twig */
            sn[0] = 0; sm[0] = (((mask_t)1) << gs) - 1; st[0] = 0; sp = 1;
            while (sp > 0) {
                sp--;
                int32_t node = sn[sp];
                mask_t mask = sm[sp];
                int under_twig = st[sp];
                int a = 0;
                mask_t open = 0;
                for (int l = 0; l < gs; l++) {
                    if (!((mask >> l) & 1)) continue;
                    a++;
                    if (is_leaf[node]) continue;
                    int64_t i = order[lo + l];
                    double dx = com[3 * node] - pos[3 * i], dy = com[3 * node + 1] - pos[3 * i + 1], dz = com[3 * node + 2] - pos[3 * i + 2];
                    double dist = sqrt(dx * dx + dy * dy + dz * dz + eps2);
                    if (!(half[node] * 2.0 / dist < theta)) open |= ((mask_t)1) << l;
                }
                if (is_leaf[node]) { q[1]++; q[4] += a; if (under_twig) { q[2]++; q[5] += a; } continue; }
                q[0]++; q[3] += a;
                int twig = 1;
                for (int c = 0; c < 8; c++) { int32_t ch = children[8 * (int64_t)node + c]; if (ch >= 0 && !is_leaf[ch]) twig = 0; }
                q[7] += twig;
                if (!open) continue;
                q[6] += twig;
                for (int c = 0; c < 8; c++) {
                    int32_t ch = children[8 * (int64_t)node + c];
                    if (ch >= 0) {
                        if (sp == cap) { cap *= 2; sn = realloc(sn, sizeof(int32_t) * cap); sm = realloc(sm, sizeof(mask_t) * cap); st = realloc(st, cap); }
                        sn[sp] = ch; sm[sp] = open; st[sp] = (uint8_t)twig; sp++;
                    }
                }
            }
            free(sn); free(sm); free(st);
        }
#pragma omp critical
        for (int k = 0; k < 8; k++) o[k] += q[k];
    }
    for (int k = 0; k < 8; k++) out[k] = o[k];
}
