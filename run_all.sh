#!/bin/bash
# Linux twin of the reference's inference/run_all.bat: one QCNN_GPU run per HEVC CTC sequence,
# same positional arguments (<ori.yuv> <anchor_prefix> <height> <width>), same order, so the 18
# doubles appended to recon_psnr.data line up with psnr_static_%d.data.
#   DATA=<dir holding HEVC_Sequence/ and anchor16.0/>  QCNN_QPS=37  QCNN_MODEL=model/qvrcnn_nchw_vect_c_8bit_qfp_%d.data  ./run_all.sh
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
EXE="${EXE:-$HERE/qcnn_gpu_amd/bin/QCNN_GPU}"
DATA="${DATA:-../../data}"
S="$DATA/HEVC_Sequence"
A="$DATA/anchor16.0"
run() { "$EXE" "$S/$1" "$A/$2" "$3" "$4"; }
run Traffic_2560x1600_30_crop10.yuv                   Traffic_intra_main_HM16.0_anchor_          1600 2560
run PeopleOnStreet_3840x2160_30_420_08_150_crop10.yuv PeopleOnStreet_intra_main_HM16.0_anchor_   2160 3840
run Kimono1_1920x1080_24_crop10.yuv                   Kimono_intra_main_HM16.0_anchor_           1080 1920
run ParkScene_1920x1080_24_crop10.yuv                 ParkScene_intra_main_HM16.0_anchor_        1080 1920
run Cactus_1920x1080_50_crop10.yuv                    Cactus_intra_main_HM16.0_anchor_           1080 1920
run BasketballDrive_1920x1080_10.yuv                  BasketballDrive_intra_main_HM16.0_anchor_  1080 1920
run BQTerrace_1920x1080_60_10.yuv                     BQTerrace_intra_main_HM16.0_anchor_        1080 1920
run BasketballDrill_832x480_50.yuv                    BasketballDrill_intra_main_HM16.0_anchor_   480  832
run BQMall_832x480_60_crop10.yuv                      BQMall_intra_main_HM16.0_anchor_            480  832
run PartyScene_832x480_50_crop10.yuv                  PartyScene_intra_main_HM16.0_anchor_        480  832
run RaceHorses_832x480_30_crop10.yuv                  RaceHorsesC_intra_main_HM16.0_anchor_       480  832
run BasketballPass_416x240_50_crop10.yuv              BasketballPass_intra_main_HM16.0_anchor_    240  416
run BQSquare_416x240_60.yuv                           BQSquare_intra_main_HM16.0_anchor_          240  416
run BlowingBubbles_416x240_50.yuv                     BlowingBubbles_intra_main_HM16.0_anchor_    240  416
run RaceHorses_416x240_30_crop10.yuv                  RaceHorses_intra_main_HM16.0_anchor_        240  416
run FourPeople_1280x720_60.yuv                        FourPeople_intra_main_HM16.0_anchor_        720 1280
run Johnny_1280x720_60_crop10.yuv                     Johnny_intra_main_HM16.0_anchor_            720 1280
run KristenAndSara_1280x720_60_crop10.yuv                 KristenAndSara_intra_main_HM16.0_anchor_    720 1280
