// ingest.cpp -- .bed -> HBM at scale.
//
// Reference: Bayes::load_genotype (src/bayes.cpp:867-900) reads this rank's marker block with one
// MPI_File_read_at per 2 GiB chunk (src/utilities.hpp:28-53) on a single thread and never looks
// at the three magic bytes.  Here: the file is validated (PLINK SNP-major magic 6c 1b 01, size
// against N and the marker range), then streamed through a small ring of pinned buffers -- a pool
// of reader threads fills chunk k+1 with parallel pread()s while the copy engine moves chunk k
// (hipMemcpy2DAsync: file rows of ceil(N/4) bytes into the 16-byte padded device stride).
#include "../../include/gmrm_hip.h"
#include "gm_host.h"
#include "gm_internal.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

using gm::fail;

#define HIPCHK_I(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = fail(GMRM_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); goto done; } } while (0)

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// pread [off, off+len) of fd into dst with nthreads concurrent readers; returns false on a short file / error
bool parallel_pread(int fd, uint8_t* dst, size_t len, off_t off, int nthreads) {
    if (len == 0) return true;
    nthreads = (int)std::max<size_t>(1, std::min<size_t>((size_t)nthreads, (len + (1u << 20) - 1) >> 20));   // >= 1 MiB per reader
    std::atomic<bool> ok{true};
    auto work = [&](int t) {
        const size_t per = (len + (size_t)nthreads - 1) / (size_t)nthreads;
        size_t b = std::min(len, per * (size_t)t), e = std::min(len, b + per);
        while (b < e) {
            const ssize_t r = pread(fd, dst + b, e - b, off + (off_t)b);
            if (r <= 0) { ok = false; return; }
            b += (size_t)r;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
    return ok.load();
}
}  // namespace

extern "C" int gmrm_load_bed_file(gmrm_ctx* c, const char* path, size_t file_first_marker, int nthreads,
                                  gmrm_ingest_stats* out) {
    if (!c || !path) return fail(GMRM_EINVAL, "null argument");
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    int rc = GMRM_OK;
    constexpr int NBUF = 3;
    uint8_t* buf[NBUF] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NBUF] = {nullptr, nullptr, nullptr};
    bool used[NBUF] = {false, false, false};
    hipStream_t st = nullptr;
    const size_t mbytes = c->mbytes, M = (size_t)c->M;
    const size_t chunk_markers = std::max<size_t>(1, ((size_t)64 << 20) / std::max<size_t>(1, mbytes));
    double t_read = 0.0;
    const double t0 = now_s();
    struct stat sb;
    unsigned char magic[3];

    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(GMRM_EIO, std::string("cannot open bed file ") + path);
    if (fstat(fd, &sb) != 0) { rc = fail(GMRM_EIO, std::string("cannot stat ") + path); goto done; }
    if (pread(fd, magic, 3, 0) != 3 || magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01) {
        rc = fail(GMRM_EIO, std::string(path) + " is not a SNP-major PLINK .bed file (magic bytes 6c 1b 01 expected)");
        goto done;
    }
    if ((size_t)sb.st_size < 3 + (file_first_marker + M) * mbytes) {
        rc = fail(GMRM_EIO, std::string(path) + ": file is shorter than 3 + (first + M) * ceil(N/4) bytes");
        goto done;
    }
    HIPCHK_I(hipSetDevice(c->device));
    HIPCHK_I(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < NBUF; i++) {
        HIPCHK_I(hipHostMalloc(reinterpret_cast<void**>(&buf[i]), chunk_markers * mbytes, hipHostMallocDefault));
        HIPCHK_I(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    {
        int k = 0;
        for (size_t m0 = 0; m0 < M; m0 += chunk_markers, k++) {
            const int b = k % NBUF;
            const size_t nm = std::min(chunk_markers, M - m0);
            if (used[b]) HIPCHK_I(hipEventSynchronize(ev[b]));           // the copy that last used this buffer is done
            const double tr0 = now_s();
            if (!parallel_pread(fd, buf[b], nm * mbytes, (off_t)(3 + (file_first_marker + m0) * mbytes), nthreads)) {
                rc = fail(GMRM_EIO, std::string(path) + ": read error / unexpected end of file");
                goto done;
            }
            t_read += now_s() - tr0;
            HIPCHK_I(hipMemcpy2DAsync(c->bed + m0 * c->stride, c->stride, buf[b], mbytes, mbytes, nm, hipMemcpyHostToDevice, st));
            HIPCHK_I(hipEventRecord(ev[b], st));
            HIPCHK_I(gm::launch_recode(c->bed + m0 * c->stride, nm * c->stride, 0, st));   // .bed code -> device code (gm_common.h), behind the copy
            used[b] = true;
        }
        HIPCHK_I(hipStreamSynchronize(st));
    }
    c->have_bed = true;
    for (auto& tr : c->tr) tr.have_stats = false;
    if (out) {
        out->bytes = M * mbytes;
        out->seconds = now_s() - t0;
        out->read_seconds = t_read;
        out->threads = nthreads;
        out->chunk_bytes = chunk_markers * mbytes;
    }
done:
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int i = 0; i < NBUF; i++) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        if (buf[i]) (void)hipHostFree(buf[i]);
    }
    close(fd);
    return rc;
}
