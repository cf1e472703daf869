/*
 * c_client.c — the drop-in boundary from PLAIN C (C99): includes include/vittrack_hip.h as a foreign
 * binding generator would (bindgen for the reference's Rust host, cgo, ...), loads
 * libvittrack_hip.so at run time and drives the reference's call sequence
 *     VitTrack::new -> init -> update ...      (src/tracker_context.rs:21,88,90,120)
 * on a raw NV12 clip. Replay harness and ABI check, not product code.
 *
 *   c_client sizes                                   print the struct sizes the header gives a C compiler
 *   c_client run <lib.so> <weights.vtw> <clip.nv12> <w> <h> <frames> <x> <y> <bw> <bh>
 *       prints one line per update: "t success score x y w h"
 *   c_client threads <same arguments>
 *       the threading row of the boundary: the reference constructs the tracker
 *       on the main thread (src/main.rs:49 -> src/pipeline_ir.rs:89) and calls it only from the GStreamer
 *       streaming thread (src/pipeline.rs:55-67,110-119). Two trackers are created on the main thread and
 *       each is driven (init + updates) by a pthread of its own, both at once; prints "k t success score
 *       x y w h" for tracker k = 0, 1 - every line must equal the single-threaded `run` output.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>
#include "../include/vittrack_hip.h"

typedef int (*fn_create)(const char*, int, const vt_config*, vt_tracker**);
typedef int (*fn_init_nv12)(vt_tracker*, const uint8_t*, const uint8_t*, int, int, int, int, vt_bbox);
typedef int (*fn_update_nv12)(vt_tracker*, const uint8_t*, const uint8_t*, int, int, int, int, vt_result*);
typedef void (*fn_destroy)(vt_tracker*);
typedef const char* (*fn_last_error)(void);
typedef int (*fn_abi)(void);
typedef void (*fn_cfg_default)(vt_config*);

typedef struct worker {
    vt_tracker* trk;
    fn_init_nv12 init_nv12;
    fn_update_nv12 update_nv12;
    fn_last_error last_error;
    const uint8_t* buf;
    size_t fbytes;
    int w, h, frames, uvs, rc;
    vt_bbox box;
    vt_result* res;      /* [frames] */
} worker;

static void* worker_main(void* p) {
    worker* k = (worker*)p;
    k->rc = k->init_nv12(k->trk, k->buf, k->buf + (size_t)k->w * k->h, k->w, k->h, k->w, k->uvs, k->box);
    for (int t = 0; t < k->frames && k->rc == VT_OK; ++t) {
        const uint8_t* y = k->buf + (size_t)t * k->fbytes;
        k->rc = k->update_nv12(k->trk, y, y + (size_t)k->w * k->h, k->w, k->h, k->w, k->uvs, &k->res[t]);
    }
    if (k->rc != VT_OK) fprintf(stderr, "worker: %d %s\n", k->rc, k->last_error());
    return NULL;
}

static void* must_sym(void* lib, const char* name) {
    void* p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "missing symbol %s\n", name);
        exit(2);
    }
    return p;
}

static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "sizes") == 0) {
        printf("vt_bbox %zu vt_result %zu vt_config %zu vt_model_info %zu vt_frame %zu abi %d max_streams %d\n",
               sizeof(vt_bbox), sizeof(vt_result), sizeof(vt_config), sizeof(vt_model_info), sizeof(vt_frame),
               VT_ABI_VERSION, VT_MAX_STREAMS);
        return 0;
    }
    if (argc != 12 || (strcmp(argv[1], "run") != 0 && strcmp(argv[1], "threads") != 0)) {
        fprintf(stderr, "usage: %s sizes | run|threads <lib> <weights> <clip.nv12> <w> <h> <frames> <x> <y> <bw> <bh>\n", argv[0]);
        return 2;
    }
    const int threaded = strcmp(argv[1], "threads") == 0;
    const char *libp = argv[2], *weights = argv[3], *clip = argv[4];
    const int w = atoi(argv[5]), h = atoi(argv[6]), frames = atoi(argv[7]);
    vt_bbox box;
    box.x = atoi(argv[8]); box.y = atoi(argv[9]); box.width = atoi(argv[10]); box.height = atoi(argv[11]);

    void* lib = dlopen(libp, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    fn_abi abi;
    fn_cfg_default cfg_default;
    fn_create create;
    fn_init_nv12 init_nv12;
    fn_update_nv12 update_nv12;
    fn_destroy destroy;
    fn_last_error last_error;
    /* POSIX form of the object-pointer -> function-pointer conversion dlsym needs */
#define LOAD(var, name) (*(void**)(&(var)) = must_sym(lib, name))
    LOAD(abi, "vt_abi_version");
    LOAD(cfg_default, "vt_config_default");
    LOAD(create, "vt_create");
    LOAD(init_nv12, "vt_init_nv12");
    LOAD(update_nv12, "vt_update_nv12");
    LOAD(destroy, "vt_destroy");
    LOAD(last_error, "vt_last_error");
#undef LOAD
    if (abi() != VT_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header %d\n", abi(), VT_ABI_VERSION);
        return 2;
    }

    const size_t fbytes = (size_t)w * h + (size_t)((w + 1) / 2 * 2) * ((h + 1) / 2);
    uint8_t* buf = (uint8_t*)malloc(fbytes * (size_t)frames);
    FILE* f = fopen(clip, "rb");
    if (!buf || !f || fread(buf, fbytes, (size_t)frames, f) != (size_t)frames) {
        fprintf(stderr, "cannot read %d frames of %zu bytes from %s\n", frames, fbytes, clip);
        return 2;
    }
    fclose(f);

    vt_config cfg;
    cfg_default(&cfg);
    vt_tracker* trk = NULL;
    int rc = create(weights, 0, &cfg, &trk);                       /* VitTrack::new */
    if (rc != VT_OK) {
        fprintf(stderr, "vt_create: %d %s\n", rc, last_error());
        return 1;
    }
    const int uvs = (w + 1) / 2 * 2;
    if (threaded) {
        vt_tracker* trk2 = NULL;
        rc = create(weights, 0, &cfg, &trk2);                      /* both constructed on the main thread */
        if (rc != VT_OK) {
            fprintf(stderr, "vt_create (2): %d %s\n", rc, last_error());
            return 1;
        }
        worker wk[2];
        pthread_t th[2];
        for (int k = 0; k < 2; ++k) {
            wk[k].trk = k ? trk2 : trk;
            wk[k].init_nv12 = init_nv12; wk[k].update_nv12 = update_nv12; wk[k].last_error = last_error;
            wk[k].buf = buf; wk[k].fbytes = fbytes; wk[k].w = w; wk[k].h = h; wk[k].frames = frames;
            wk[k].uvs = uvs; wk[k].rc = 0; wk[k].box = box;
            wk[k].res = (vt_result*)calloc((size_t)frames, sizeof(vt_result));
            if (!wk[k].res || pthread_create(&th[k], NULL, worker_main, &wk[k]) != 0) {
                fprintf(stderr, "cannot start worker %d\n", k);
                return 1;
            }
        }
        for (int k = 0; k < 2; ++k) pthread_join(th[k], NULL);
        for (int k = 0; k < 2; ++k) {
            if (wk[k].rc != VT_OK) return 1;
            for (int t = 0; t < frames; ++t) {
                const vt_result* r = &wk[k].res[t];
                printf("%d %d %d %.9g %d %d %d %d\n", k, t, r->success, r->score, r->bbox.x, r->bbox.y, r->bbox.width,
                       r->bbox.height);
            }
            free(wk[k].res);
        }
        destroy(trk);                                              /* and destroyed on the main thread */
        destroy(trk2);
        free(buf);
        dlclose(lib);
        return 0;
    }
    rc = init_nv12(trk, buf, buf + (size_t)w * h, w, h, w, uvs, box);   /* tracker.init(frame 0, bbox) */
    if (rc != VT_OK) {
        fprintf(stderr, "vt_init_nv12: %d %s\n", rc, last_error());
        return 1;
    }
    double* lat = (double*)calloc((size_t)frames, sizeof(double));
    for (int t = 0; t < frames; ++t) {                              /* tracker.update(frame t) */
        const uint8_t* y = buf + (size_t)t * fbytes;
        vt_result r;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        rc = update_nv12(trk, y, y + (size_t)w * h, w, h, w, uvs, &r);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc != VT_OK) {
            fprintf(stderr, "vt_update_nv12: %d %s\n", rc, last_error());
            return 1;
        }
        if (lat) lat[t] = (double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3;
        printf("%d %d %.9g %d %d %d %d\n", t, r.success, r.score, r.bbox.x, r.bbox.y, r.bbox.width, r.bbox.height);
    }
    if (lat && frames > 40) {       /* what a compiled host sees per update (stderr; the first 20 updates are warm-up) */
        const int n = frames - 20;
        qsort(lat + 20, (size_t)n, sizeof(double), cmp_double);
        fprintf(stderr, "update latency from C, %d updates: p50 %.1f us  p99 %.1f us\n", n, lat[20 + n / 2], lat[20 + (n * 99) / 100]);
    }
    free(lat);
    destroy(trk);
    free(buf);
    dlclose(lib);
    return 0;
}
