// Context of the MI355X WSPR decoder (one per process, created lazily on the first
// call): HIP stream, constant tables, grow-only device/pinned buffers, host pool.
#pragma once
#include <chrono>
#include <memory>
#include <vector>

#include "../../../include/wspr_mi355x.h"
#include "../kernels/wspr_device.h"

namespace wspr {

class Context {
public:
    static Context& get();          // slot 0; throws std::runtime_error when no HIP device is usable
    static Context& slot(int i);    // i in [0, slots())
    static int slots();             // concurrent pipelines per process (env WSPR_SLOTS, default 3)
    int device();
    ~Context();

    hipStream_t stream();
    const DeviceTables& tables();
    int host_threads();

    // working IQ buffers (planar, rows of kIqStride floats)
    float* work_i(int nseg);
    float* work_q(int nseg);
    void load_host(const float* I, const float* Q, int nseg, int samples, size_t stride);
    void load_device(const void* dI, const void* dQ, int nseg, int samples, size_t stride);
    void store_host(float* I, float* Q, int nseg, int samples, size_t stride);
    void sync();

    float* ps_buffer(int nseg);
    void run_fft_sync(int nseg, int samples, int maxdrift, bool coarse, const int* d_seglist, int nactive,
                      float* noise_out, float* smspec_out);
    void fetch_candidates(int nseg, std::vector<int>& npk, std::vector<DevCand>& cand);

    // the decoder proper, on the working buffers
    int decode_resident(int nseg, int samples, const decoder_options& opt, decoder_results* out,
                        int max_results, int* n_results);
    int last_timings(double* ms, int cap);
    int bench_fft_sync(int nseg, int samples, int iters, double* ms);

    void demod_single(float* id, float* qd, long np, unsigned char* symbols, float* freq, int ifmin, int ifmax,
                      float fstep, int* shift, int lagmin, int lagmax, int lagstep, float* drift, float* sync,
                      int mode);
    void subtract_single(float* id, float* qd, long np, float f0, int shift, float drift, const unsigned char* sym);
    int bench_decimate(const void* d_raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ, int iters, double* ms);
    int decimate_device(const void* d_raw, size_t bytes_per_seg, int nseg, float* dI, float* dQ, int normalise,
                        int* h_nout);

    struct Impl;
    std::unique_ptr<Impl> d;

private:
    explicit Context(int nslots);
};

}  // namespace wspr
