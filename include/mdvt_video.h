/*
 * mdvt_video.h -- C ABI of libmdvt_video.so: the on-disk format either side of the stereo-rerender path
 * (SURVEY.md 8f rank 2): FFV1 video in a Matroska container, as OpenCV's VideoWriter / VideoCapture produce and consume it
 * for the reference.  Host code only (C++17, no GPU, no third-party library): FFmpeg / OpenCV are absent from the image,
 * so both directions are written from the published specifications -- FFV1: RFC 9043 (versions 0, 1 and 3; range coder with
 * the default or a custom state-transition table and Golomb-Rice; the JPEG 2000 RCT "RGB" colour space at 8 bits with or
 * without alpha; slices with CRC-32 parities); Matroska: RFC 9559 / EBML RFC 8794 (the subset one video track needs).
 *
 * What it replaces in the reference:
 *   reader  cv2.VideoCapture(depth_video / color_video) + .read()             stereo_rerender.py:326-341, 489-509
 *           (the toolbox's own *_depth.mkv files; a colour video only if it is FFV1-in-Matroska too: H.264 etc. are not decoded)
 *   writer  cv2.VideoWriter(path, fourcc('F','F','V','1'), fps, (w, h)) + .write()   stereo_rerender.py:426-444, 941;
 *                                                                              depth_frames_helper.py:125-161
 * INTEROPERABILITY UNPINNED: no FFmpeg exists here to read these files or to produce files for the reader.  The encoder is
 * checked by an independent decoder restated from the RFC's pseudo-code (test infrastructure: ffv1_ref.py next to the C oracle), the container by structural
 * tests (EBML sizes, CRCs); tests/golden/gen_ffv1_golden.py produces cross-check vectors on a machine that has ffmpeg.
 */
#ifndef MDVT_VIDEO_H
#define MDVT_VIDEO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDVT_VIDEO_ABI 1

typedef struct mdvt_video_reader mdvt_video_reader;
typedef struct mdvt_video_writer mdvt_video_writer;

typedef struct mdvt_video_info {
    int32_t width, height;
    int64_t frames;              /* video frames indexed in the file                                   */
    double fps;                  /* from the track's DefaultDuration (or the Info duration / frames)   */
    int32_t ffv1_version;        /* 0, 1 or 3                                                          */
    int32_t ffv1_micro_version;
    int32_t coder_type;          /* 0 Golomb-Rice, 1 range coder default table, 2 range coder custom   */
    int32_t slices;              /* per frame (1 for versions 0 / 1)                                   */
    int32_t alpha;               /* the stream carries a fourth plane (ignored on output)              */
    int32_t intra;               /* every frame is a key frame                                         */
    int32_t ec;                  /* slices end in a CRC-32 parity                                      */
    int32_t reserved;
} mdvt_video_info;

/* Pixel order of the caller's interleaved 8-bit buffers. */
enum { MDVT_VIDEO_RGB = 0, MDVT_VIDEO_BGR = 1 };   /* BGR = what cv2 hands the reference (sr:489-509 convert it) */

/* 0 on success; a negative code otherwise, with a message in mdvt_video_last_error() (thread local). */
const char* mdvt_video_last_error(void);
int mdvt_video_abi(void);

/* Opens a Matroska file, finds its (first) FFV1 video track, parses the configuration and indexes the frames. */
int mdvt_video_open(const char* path, mdvt_video_reader** out, mdvt_video_info* info);
/* Decodes the NEXT frame (streams with inter-frame context state must be read in order) into dst: height rows of 3*width
 * bytes, `pitch` bytes apart.  threads <= 0: one per slice, capped at the host's cores.  Returns 1 at the end of the video. */
int mdvt_video_read(mdvt_video_reader* r, uint8_t* dst, size_t pitch, int order, int threads);
/* Restarts at frame 0 (a key frame). */
int mdvt_video_rewind(mdvt_video_reader* r);
/* Makes `frame` the next one mdvt_video_read returns.  Free for an intra-only stream; otherwise the frames from the last key
 * frame at or before it are decoded and dropped (a key frame is a packet whose first range-coded bit is 1).  frame == frames:
 * the end. */
int mdvt_video_seek(mdvt_video_reader* r, int64_t frame, int threads);
/* The next frame's FFV1 packet as stored, without decoding it (remuxing); advances like mdvt_video_read.  1 at the end. */
int mdvt_video_next_packet(mdvt_video_reader* r, uint8_t* packet, size_t packet_cap, size_t* packet_size);
/* The stream's configuration record (empty for versions 0 / 1). */
int mdvt_video_config_record(mdvt_video_reader* r, uint8_t* config, size_t config_cap, size_t* config_size);
void mdvt_video_close(mdvt_video_reader* r);

/* Creates path and writes the headers.  FFV1 version 3, range coder (default table), intra-only, RGB colour space, 8 bits,
 * slices_h x slices_v slices (each >= 1, product <= 1024; 0, 0 = 4 x 4), CRC-32 parities.  fps as a rational. */
int mdvt_video_create(const char* path, int width, int height, int fps_num, int fps_den, int slices_h, int slices_v,
                      mdvt_video_writer** out);
/* Encodes one frame (threads as above) and appends it. */
int mdvt_video_write(mdvt_video_writer* w, const uint8_t* src, size_t pitch, int order, int threads);
/* Appends a frame already encoded by mdvt_ffv1_encode_frame with this writer's width, height and slice counts (frames encoded in
 * parallel by the caller, or taken from another file of the same configuration with mdvt_video_next_packet). */
int mdvt_video_write_packet(mdvt_video_writer* w, const uint8_t* packet, size_t packet_size);
/* Writes the cues, patches the duration and the segment size, closes the file; *frames = frames written.  The writer is freed. */
int mdvt_video_finish(mdvt_video_writer* w, int64_t* frames);

/* The codec alone (tests, the independent decoder's counterpart): one frame <-> one FFV1 packet + the configuration record. */
int mdvt_ffv1_encode_frame(int width, int height, int slices_h, int slices_v, const uint8_t* src, size_t pitch, int order,
                           int threads, uint8_t* packet, size_t packet_cap, size_t* packet_size,
                           uint8_t* config, size_t config_cap, size_t* config_size);

#ifdef __cplusplus
}
#endif
#endif
