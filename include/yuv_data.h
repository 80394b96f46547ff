/*
 * yuv_data.h -- the reference's YUV / PSNR helper (inference/yuv_data.h:11-27): planar 4:2:0
 * 8-bit files, luma only, whole-clip and per-frame PSNR, reconstruction writer.
 */
#ifndef QCNN_YUV_DATA_H
#define QCNN_YUV_DATA_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mat.h"

class vrcnn_data {
public:
    vrcnn_data(int frame, int height, int width);
    int read_data(char *orifile, char *inputfile);          /* inference/yuv_data.cpp:15-42 */
    int read_frame(char *orifile, char *inputfile, int n);  /* inference/yuv_data.cpp:44-66 */
    double psnr(datatype *data);                            /* inference/yuv_data.cpp:87-97 */
    double psnr_pf(void);                                   /* inference/yuv_data.cpp:98-112 */
    double psnr_frame(const datatype *data, int n);         /* PSNR of frame n of `data` */
    int save_recon_as(char *filename);                      /* inference/yuv_data.cpp:113-128 */
    /* Chroma extension (the reference skips the chroma bytes, inference/yuv_data.cpp:36-37, and writes zeros
     * for them, :119-125).  On an object built as (2*frames, height/2, width/2): reads the U and V planes of
     * every frame of both files as planes 2i and 2i+1; patch_chroma_into() overwrites the zero chroma of a
     * file written by save_recon_as with this object's recon planes. */
    int read_chroma(char *orifile, char *inputfile);
    int patch_chroma_into(char *filename);
    static double psnr_from_sse(unsigned long long sse, long n); /* 10*log10(65025 / (sse/n)), :94-96 */
    ~vrcnn_data(void);

    int frame, h, w;
    long nSize;
    datatype *ori, *input, *recon;
    bool pinned; /* the three buffers are page-locked (qcnn_host_alloc_pinned) */
};

/*
 * Streaming twin of vrcnn_data::read_data / save_recon_as (inference/yuv_data.cpp:15-42,113-128) for clips that
 * should not sit in host memory whole: the same files, the same bytes, any range of frames at a time.  Frame i of
 * a 4:2:0 file starts at byte i * h*w*3/2, so ranges can be read and written by several threads at once (every
 * call is one positioned read or write per frame; the object keeps no file position).  The recon file is sized
 * to frames * h*w*3/2 when it is opened: the chroma bytes the reference writes as zeros
 * (inference/yuv_data.cpp:119-125) are the zeros a file is extended with, only luma is written.
 * Errors behave like read_data: a message on stdout and exit(1).
 */
class vrcnn_stream {
public:
    vrcnn_stream(char *orifile, char *inputfile, int frame, int height, int width);
    /* luma of frames first .. first+count-1 of both files into ori / input (count * h*w bytes each) */
    void read_frames(int first, int count, datatype *ori, datatype *input) const;
    void open_recon(char *filename);
    void write_frames(int first, int count, const datatype *recon) const;
    ~vrcnn_stream(void);

    int frame, h, w;
    long nSize; /* frame * h * w, as vrcnn_data::nSize */

private:
    int ori_fd, input_fd, recon_fd;
};

#endif
