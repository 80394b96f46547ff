// vrcnn_data of the reference (inference/yuv_data.cpp): YUV 4:2:0 luma reader, PSNR, recon writer.
#include "yuv_data.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

vrcnn_data::vrcnn_data(int frame, int height, int width)
{
    this->frame = frame;
    this->h = height;
    this->w = width;
    this->nSize = (long)frame * height * width;
    // Page-locked host memory when the runtime grants it (the frame copies are then true DMA and the
    // pipelined clip entry point can overlap them with the kernels), ordinary memory otherwise.
    void *p[3] = {nullptr, nullptr, nullptr};
    pinned = true;
    for (int i = 0; i < 3 && pinned; ++i) pinned = qcnn_host_alloc_pinned(&p[i], (size_t)nSize) == 0;
    if (pinned) {
        this->ori = static_cast<datatype *>(p[0]);
        this->input = static_cast<datatype *>(p[1]);
        this->recon = static_cast<datatype *>(p[2]);
    } else {
        for (int i = 0; i < 3; ++i)
            if (p[i]) qcnn_host_free_pinned(p[i]);
        this->ori = new datatype[nSize];
        this->input = new datatype[nSize];
        this->recon = new datatype[nSize];
    }
}

int vrcnn_data::read_data(char *orifile, char *inputfile)
{
    FILE *orifp = fopen(orifile, "rb"), *inputfp;
    if (!orifp) {
        printf("%s\n", orifile);
        printf("open ori file failed\n");
        exit(1);
    }
    inputfp = fopen(inputfile, "rb");
    if (!inputfp) {
        printf("%s\n", inputfile);
        printf("open input file failed\n");
        fclose(orifp);
        exit(1);
    }
    const long hw = (long)h * w;
    for (int i = 0; i < frame; ++i) {
        // luma, then skip the h*w/2 chroma bytes (inference/yuv_data.cpp:34-37)
        if (fread(ori + i * hw, sizeof(datatype), hw, orifp) != (size_t)hw ||
            fread(input + i * hw, sizeof(datatype), hw, inputfp) != (size_t)hw) {
            printf("short read at frame %d\n", i);
            exit(1);
        }
        fseek(orifp, sizeof(datatype) * hw / 2, SEEK_CUR);
        fseek(inputfp, sizeof(datatype) * hw / 2, SEEK_CUR);
    }
    fclose(orifp);
    fclose(inputfp);
    return 0;
}

int vrcnn_data::read_chroma(char *orifile, char *inputfile)
{
    FILE *orifp = fopen(orifile, "rb"), *inputfp;
    if (!orifp) {
        printf("%s\n", orifile);
        printf("open ori file failed\n");
        exit(1);
    }
    inputfp = fopen(inputfile, "rb");
    if (!inputfp) {
        printf("%s\n", inputfile);
        printf("open input file failed\n");
        fclose(orifp);
        exit(1);
    }
    const long q = (long)h * w;  // one chroma plane; the luma plane of the frame holds 4q bytes
    for (int i = 0; i < frame / 2; ++i) {
        fseek(orifp, sizeof(datatype) * 4 * q, SEEK_CUR);
        fseek(inputfp, sizeof(datatype) * 4 * q, SEEK_CUR);
        if (fread(ori + 2 * i * q, sizeof(datatype), 2 * q, orifp) != (size_t)(2 * q) ||
            fread(input + 2 * i * q, sizeof(datatype), 2 * q, inputfp) != (size_t)(2 * q)) {
            printf("short read at frame %d\n", i);
            exit(1);
        }
    }
    fclose(orifp);
    fclose(inputfp);
    return 0;
}

int vrcnn_data::patch_chroma_into(char *filename)
{
    FILE *fp = fopen(filename, "r+b");
    if (!fp) {
        printf("write file failed\n");
        return 1;
    }
    const long q = (long)h * w;
    for (int i = 0; i < frame / 2; ++i) {
        fseek(fp, (long)i * 6 * q + 4 * q, SEEK_SET);
        fwrite(recon + 2 * i * q, sizeof(datatype), 2 * q, fp);
    }
    fclose(fp);
    return 0;
}

int vrcnn_data::read_frame(char *orifile, char *inputfile, int n)
{
    FILE *orifp = fopen(orifile, "rb"), *inputfp;
    frame = 1;
    if (!orifp) {
        printf("open file failed\n");
        return 1;
    }
    inputfp = fopen(inputfile, "rb");
    if (!inputfp) {
        printf("open file failed\n");
        fclose(orifp);
        return 1;
    }
    const long hw = (long)h * w;
    fseek(orifp, sizeof(datatype) * hw * n * 3 / 2, SEEK_CUR);
    size_t got = fread(ori, sizeof(datatype), hw, orifp);
    fseek(inputfp, sizeof(datatype) * hw * n * 3 / 2, SEEK_CUR);
    got += fread(input, sizeof(datatype), hw, inputfp);
    fclose(orifp);
    fclose(inputfp);
    return got == (size_t)(2 * hw) ? 0 : 1;
}

static double psnr_of(const datatype *data, const datatype *ori, long n)
{
    // inference/yuv_data.cpp:91-95; integer accumulation gives the same double (sums < 2^53)
    unsigned long long sse = 0;
    for (long i = 0; i < n; i++) {
        const int d = (int)data[i] - (int)ori[i];
        sse += (unsigned)(d * d);
    }
    double mse = (double)sse;
    mse /= (double)n;
    return 10 * log10(65025.0 / mse);
}

double vrcnn_data::psnr(datatype *data) { return psnr_of(data, ori, nSize); }

double vrcnn_data::psnr_frame(const datatype *data, int n)
{
    const long hw = (long)h * w;
    return psnr_of(data + n * hw, ori + n * hw, hw);
}

double vrcnn_data::psnr_pf(void)
{
    double p = 0;
    for (int n = 0; n < frame; n++) {
        p = psnr_frame(recon, n);
        printf("PSNR of Frame %d:%f\n", n, p);  // inference/yuv_data.cpp:109
    }
    return p;
}

int vrcnn_data::save_recon_as(char *filename)
{
    FILE *fp = fopen(filename, "wb");
    if (!fp) {
        printf("write file failed\n");
        return 1;
    }
    const long hw = (long)h * w;
    datatype *uv = new datatype[hw / 2];
    memset(uv, 0, hw / 2);
    for (int i = 0; i < frame; i++) {
        fwrite(recon + i * hw, sizeof(datatype), hw, fp);
        fwrite(uv, sizeof(datatype), hw / 2, fp);
    }
    delete[] uv;
    fclose(fp);
    return 0;
}

vrcnn_data::~vrcnn_data(void)
{
    if (pinned) {
        qcnn_host_free_pinned(ori);
        qcnn_host_free_pinned(input);
        qcnn_host_free_pinned(recon);
    } else {
        delete[] ori;
        delete[] input;
        delete[] recon;
    }
}

double vrcnn_data::psnr_from_sse(unsigned long long sse, long n)
{
    double mse = (double)sse;
    mse /= (double)n;
    return 10 * log10(65025.0 / mse);
}

// ---------------------------------------------------------------------------------------------
// vrcnn_stream: positioned reads / writes of frame ranges (see include/yuv_data.h)
vrcnn_stream::vrcnn_stream(char *orifile, char *inputfile, int frame_, int height, int width)
    : frame(frame_), h(height), w(width), nSize((long)frame_ * height * width), ori_fd(-1), input_fd(-1), recon_fd(-1)
{
    ori_fd = open(orifile, O_RDONLY);
    if (ori_fd < 0) {
        printf("%s\n", orifile);
        printf("open ori file failed\n");  // inference/yuv_data.cpp:19-23
        exit(1);
    }
    input_fd = open(inputfile, O_RDONLY);
    if (input_fd < 0) {
        printf("%s\n", inputfile);
        printf("open input file failed\n");  // inference/yuv_data.cpp:25-31
        exit(1);
    }
}

// Leaves the process the way read_data does (message on stdout, status 1), from whichever thread found the problem:
// the other threads of a streaming run are still using the GPU, so no destructor or atexit handler may run.
static void die(void)
{
    fflush(stdout);
    _exit(1);
}

static void read_fully(int fd, datatype *dst, size_t n, off_t at, int frame_no)
{
    while (n > 0) {
        const ssize_t got = pread(fd, dst, n, at);
        if (got <= 0) {
            printf("short read at frame %d\n", frame_no);
            die();
        }
        dst += got, at += got, n -= (size_t)got;
    }
}

void vrcnn_stream::read_frames(int first, int count, datatype *ori, datatype *input) const
{
    const size_t hw = (size_t)h * w;
    for (int i = 0; i < count; ++i) {
        // luma only; the h*w/2 chroma bytes behind it are skipped (inference/yuv_data.cpp:34-37)
        const off_t at = (off_t)(first + i) * (off_t)(hw + hw / 2);
        read_fully(ori_fd, ori + i * hw, hw, at, first + i);
        read_fully(input_fd, input + i * hw, hw, at, first + i);
    }
}

void vrcnn_stream::open_recon(char *filename)
{
    recon_fd = open(filename, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    const size_t hw = (size_t)h * w;
    if (recon_fd < 0 || ftruncate(recon_fd, (off_t)frame * (off_t)(hw + hw / 2)) != 0) {
        printf("write file failed\n");  // inference/yuv_data.cpp:116
        exit(1);
    }
}

void vrcnn_stream::write_frames(int first, int count, const datatype *recon) const
{
    if (recon_fd < 0) return;
    const size_t hw = (size_t)h * w;
    for (int i = 0; i < count; ++i) {
        const datatype *src = recon + i * hw;
        off_t at = (off_t)(first + i) * (off_t)(hw + hw / 2);
        size_t n = hw;
        while (n > 0) {
            const ssize_t put = pwrite(recon_fd, src, n, at);
            if (put <= 0) {
                printf("write file failed\n");
                die();
            }
            src += put, at += put, n -= (size_t)put;
        }
    }
}

vrcnn_stream::~vrcnn_stream(void)
{
    if (ori_fd >= 0) close(ori_fd);
    if (input_fd >= 0) close(input_fd);
    if (recon_fd >= 0) close(recon_fd);
}
