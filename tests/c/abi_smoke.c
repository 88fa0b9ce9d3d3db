/* Plain-C consumer of include/cloops_hip.h (no Python, no C++): what a non-Python host would write.
 * Clusters a small synthetic chromosome with every variant through the C ABI and checks the table against
 * the labels.  Built and run by tests/test_gpu_c_abi.py:  gcc -std=c99 -I include abi_smoke.c -L... -lcloops_hip */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "cloops_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != CL_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cl_last_error()); return 1; } } while (0)

int main(void)
{
    const int64_t n = 200000;
    int32_t *x = malloc(n * sizeof *x), *y = malloc(n * sizeof *y), *lab = malloc(n * sizeof *lab);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        int32_t a = (int32_t)(s % 40000000u);
        int32_t anchor = (int32_t)((s >> 32) % 2000u) * 20000;           /* 2000 loop anchors */
        if (i % 3) { x[i] = anchor + (int32_t)(s % 600u); y[i] = anchor + 150000 + (int32_t)((s >> 20) % 600u); }
        else { x[i] = a; y[i] = a + (int32_t)((s >> 40) % 500000u); }
    }
    if (cl_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    cl_chrom* c = NULL;
    CHECK(cl_chrom_create(0, NULL, x, y, n, 0, &c));
    if (cl_chrom_size(c) != n) return 3;
    const int variants[3] = {CL_VARIANT_CDBSCAN2, CL_VARIANT_CDBSCAN1, CL_VARIANT_BLOCK};
    for (int v = 0; v < 3; ++v) {
        int32_t nc = 0, ml = -1;
        CHECK(cl_cluster(c, variants[v], 2000, 5, 0, lab, &nc, &ml));
        cl_box* boxes = malloc((size_t)(ml + 1) * sizeof *boxes);
        CHECK(cl_get_boxes(c, boxes));
        int64_t* cnt = calloc((size_t)(ml + 1), sizeof *cnt);
        int64_t labelled = 0;
        for (int64_t i = 0; i < n; ++i) {
            if (lab[i] < 0) continue;
            if (lab[i] > ml) { fprintf(stderr, "label beyond max_label\n"); return 4; }
            ++cnt[lab[i]]; ++labelled;
            const cl_box* b = &boxes[lab[i]];
            if (x[i] < b->min_x || x[i] > b->max_x || y[i] < b->min_y || y[i] > b->max_y) { fprintf(stderr, "PET outside its box\n"); return 5; }
        }
        int32_t live = 0;
        for (int32_t k = 0; k <= ml; ++k) {
            if (cnt[k] != boxes[k].count) { fprintf(stderr, "count mismatch at %d\n", k); return 6; }
            live += cnt[k] > 0;
        }
        if (live != nc || nc < 100) { fprintf(stderr, "cluster count %d vs %d\n", live, nc); return 7; }
        printf("variant %d: %d clusters, %lld labelled PETs\n", variants[v], nc, (long long)labelled);
        free(boxes); free(cnt);
    }
    int32_t nc = 0, ml = -1;
    CHECK(cl_cluster_weighted(c, 20000, 5, 50, 1, lab, &nc, &ml));
    printf("weighted (50,1): %d clusters\n", nc);
    {
        /* the inner loop of cLoops/pipe.py:247-250 from C: the minPts list announced, the first run keeps its neighbour counts,
           the runs that follow at another minPts / cut query their cut band only -- labels equal to a handle without the re-use */
        const int32_t served[3] = {8, 5, 3};
        const int32_t runs[4][2] = {{8, 0}, {5, 700}, {3, 1500}, {5, 0}};
        const int want_mode[4] = {0, 2, 2, 1};
        cl_chrom* d = NULL;
        CHECK(cl_chrom_create(0, NULL, x, y, n, 0, &d));
        cl_set_count_reuse(d, 0);
        cl_set_count_thresholds(c, served, 3);
        int32_t* lab2 = malloc((size_t)n * sizeof *lab2);
        if (!lab2) return 10;
        for (int r = 0; r < 4; ++r) {
            int32_t nc2 = 0, ml2 = -1;
            CHECK(cl_cluster(c, CL_VARIANT_CDBSCAN2, 2000, runs[r][0], runs[r][1], lab, &nc, &ml));
            if (cl_last_region_mode(c) != want_mode[r]) { fprintf(stderr, "run %d: region mode %d, expected %d\n", r, cl_last_region_mode(c), want_mode[r]); return 11; }
            CHECK(cl_cluster(d, CL_VARIANT_CDBSCAN2, 2000, runs[r][0], runs[r][1], lab2, &nc2, &ml2));
            if (cl_last_region_mode(d) != 0 || nc != nc2 || ml != ml2) return 12;
            for (int64_t i = 0; i < n; ++i) if (lab[i] != lab2[i]) { fprintf(stderr, "run %d: label of row %lld differs\n", r, (long long)i); return 13; }
        }
        printf("count cache: 4 runs on kept words, labels equal\n");
        free(lab2);
        cl_chrom_destroy(d);
    }
    {
        /* the outer loop of cLoops/pipe.py:241-281 from C: the eps list announced (cl_set_eps_list), both layouts merged from one fine
           sort -- labels equal to a handle that sorts every layout */
        const int32_t eps_list[2] = {1000, 2000};
        cl_chrom* d = NULL;
        CHECK(cl_chrom_create(0, NULL, x, y, n, 0, &d));
        cl_set_sort_index(c, 1);
        cl_set_eps_list(c, eps_list, 2);
        int32_t* lab2 = malloc((size_t)n * sizeof *lab2);
        if (!lab2) return 20;
        for (int r = 0; r < 2; ++r) {
            int32_t nc2 = 0, ml2 = -1;
            CHECK(cl_cluster(c, CL_VARIANT_CDBSCAN2, eps_list[r], 5, 300, lab, &nc, &ml));
            CHECK(cl_cluster(d, CL_VARIANT_CDBSCAN2, eps_list[r], 5, 300, lab2, &nc2, &ml2));
            if (nc != nc2 || ml != ml2) return 21;
            for (int64_t i = 0; i < n; ++i) if (lab[i] != lab2[i]) { fprintf(stderr, "eps %d: label of row %lld differs\n", eps_list[r], (long long)i); return 22; }
        }
        cl_set_eps_list(c, NULL, 0);
        printf("eps list: 2 layouts from one fine sort, labels equal\n");
        free(lab2);
        cl_chrom_destroy(d);
    }
    {
        /* a sweep step + the candidate table left on the device (cl_cand_finish_device: what cl_comm_gather_device sends) next to the
           host form: the same rows */
        int64_t ni = 0, ns = 0, k_host = 0, k_dev = 0;
        const int32_t* dev_rows = NULL;
        cl_dsummary* sm = malloc(sizeof *sm);
        if (!sm) return 14;
        CHECK(cl_cand_reset(c));
        CHECK(cl_cluster_step_async(c, CL_VARIANT_CDBSCAN2, 2000, 5, 0, 0, -1));
        CHECK(cl_wait(c, &nc, &ml));
        CHECK(cl_step_result(c, &ni, &ns, sm));
        int32_t* hb = malloc((size_t)(ni + 1) * 16);
        CHECK(cl_cand_finish(c, 0, hb, ni, &k_host));
        CHECK(cl_cand_finish_device(c, 0, &dev_rows, &k_dev));
        if (k_host != k_dev || (k_dev > 0 && !dev_rows)) { fprintf(stderr, "cand_finish_device: %lld rows, host form %lld\n", (long long)k_dev, (long long)k_host); return 15; }
        printf("sweep step: %lld inter-ligation boxes, %lld kept (host and device forms)\n", (long long)ni, (long long)k_dev);
        free(hb); free(sm);
    }
    {
        /* a whole sweep announced in ONE call (cl_sweep_plan): `for ep in eps: for m in minPts:` of cLoops/pipe.py:241-281 with a
           moving cut -- labels equal to a handle without any plan, and the plan is what makes the later runs of an eps cheap
           (region mode 2: words re-used, cut band re-queried) */
        const int32_t eps_list[2] = {1000, 2000}, mp_list[2] = {8, 4};
        cl_chrom* d = NULL;
        CHECK(cl_chrom_create(0, NULL, x, y, n, 0, &d));
        cl_set_count_reuse(d, 0);
        CHECK(cl_sweep_plan(c, eps_list, 2, mp_list, 2));
        int32_t* lab2 = malloc((size_t)n * sizeof *lab2);
        if (!lab2) return 30;
        int32_t cut = 0;
        for (int e = 0; e < 2; ++e) for (int m = 0; m < 2; ++m) {
            int32_t nc2 = 0, ml2 = -1;
            CHECK(cl_cluster(c, CL_VARIANT_CDBSCAN2, eps_list[e], mp_list[m], cut, lab, &nc, &ml));
            if (cl_last_region_mode(c) != (m == 0 ? 0 : 2)) { fprintf(stderr, "plan: run (%d, %d) region mode %d\n", e, m, cl_last_region_mode(c)); return 31; }
            CHECK(cl_cluster(d, CL_VARIANT_CDBSCAN2, eps_list[e], mp_list[m], cut, lab2, &nc2, &ml2));
            if (nc != nc2 || ml != ml2) return 32;
            for (int64_t i = 0; i < n; ++i) if (lab[i] != lab2[i]) { fprintf(stderr, "plan: label of row %lld differs\n", (long long)i); return 33; }
            cut += 400;
        }
        CHECK(cl_sweep_plan(c, NULL, 0, NULL, 0));
        if (cl_sweep_plan(c, eps_list, -1, mp_list, 2) != CL_ERR_ARG) return 34;
        printf("sweep plan: 4 runs, labels equal to a handle without a plan\n");
        free(lab2);
        cl_chrom_destroy(d);
    }
    {
        /* (row, label) pairs with a buffer that is too small: cl_wait refuses (CL_ERR_ARG), nothing is written beyond the capacity */
        int32_t* pin = (int32_t*)cl_host_alloc(64 * 8 + 8);
        if (!pin) return 40;
        pin[128] = 0x5a5a5a5a;
        CHECK(cl_cluster_pairs_async(c, CL_VARIANT_CDBSCAN2, 2000, 5, 0, pin, 64));
        if (cl_wait(c, &nc, &ml) != CL_ERR_ARG) { fprintf(stderr, "pairs: a 64-pair buffer was accepted\n"); return 41; }
        if (pin[128] != 0x5a5a5a5a) return 42;
        cl_host_free(pin);
        CHECK(cl_cluster(c, CL_VARIANT_CDBSCAN2, 2000, 5, 0, lab, &nc, &ml));     /* the handle is still usable */
        printf("pairs capacity: refused\n");
    }
    {
        /* the same labels as one bit per row + the labels of the set rows in row order (cl_cluster_rowmask_async) */
        const size_t nw = (size_t)((n + 63) / 64);
        unsigned char* pin = (unsigned char*)cl_host_alloc((int64_t)(8 * nw + 4 * (size_t)n));
        if (!pin) return 50;
        CHECK(cl_cluster(c, CL_VARIANT_CDBSCAN2, 2000, 5, 0, lab, &nc, &ml));
        CHECK(cl_cluster_rowmask_async(c, CL_VARIANT_CDBSCAN2, 2000, 5, 0, pin, n));
        int32_t nc3 = 0, ml3 = -1;
        CHECK(cl_wait(c, &nc3, &ml3));
        const unsigned long long* mask = (const unsigned long long*)pin;
        const int32_t* rl = (const int32_t*)(pin + 8 * nw);
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int set = (int)((mask[i >> 6] >> (i & 63)) & 1ull);
            if (set != (lab[i] >= 0)) { fprintf(stderr, "rowmask: bit of row %lld\n", (long long)i); return 51; }
            if (set && rl[k++] != lab[i]) { fprintf(stderr, "rowmask: label of row %lld\n", (long long)i); return 52; }
        }
        if (k != cl_last_n_labelled(c) || nc3 != nc) return 53;
        cl_host_free(pin);
        printf("rowmask: %lld labelled rows, bits and labels equal to cl_cluster\n", (long long)k);
    }
    if (cl_cluster(c, 7, 2000, 5, 0, lab, &nc, &ml) != CL_ERR_ARG) return 8;          /* unknown variant */
    if (cl_cluster(c, CL_VARIANT_CDBSCAN2, 0, 5, 0, lab, &nc, &ml) != CL_ERR_ARG) return 9;   /* eps = 0 */
    cl_chrom_destroy(c);
    free(x); free(y); free(lab);
    printf("abi smoke ok (library version %d)\n", cl_version());
    return 0;
}
