// jitter.hpp -- the harness's quality score.
//
// The reference scores a clip by the median, over frame pairs, of the per-pair median optical-flow magnitude, with the
// flow field from cv::calcOpticalFlowFarneback (eval_jitter.cpp:43-70, grid_search_align.cpp:27-60); the grid searches
// report output jitter / input jitter.  Farneback is OpenCV code that is not in this build.  The stand-in keeps the
// statistic and replaces the flow field by the one this library measures: the 4-parameter similarity between
// successive frames, evaluated at a lattice of pixel positions.  For camera shake (global motion) the two fields agree;
// independently moving objects, which the median suppresses in the reference too, are invisible here.  Scores are
// comparable between clips measured by this tool, not with numbers printed by the reference's binaries.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "vs_amd.h"

namespace vsjit {

// printed once by every tool that reports the score (the score must say what it is where it is read)
inline const char* score_note() {
    return "note: jitter score = median flow magnitude of the similarity THIS library measures between successive frames (not OpenCV's "
           "Farneback flow, eval_jitter.cpp:43-70): it sees global camera motion only, and a grid search over aligner parameters scores "
           "an aligner with an aligner -- failures that fool the measuring aligner as well go unseen; compare scores of this tool only";
}


// median with the mean of the two middle elements for even sizes (eval_jitter.cpp:8-19)
inline double median(std::vector<double>& v) {
    if (v.empty()) return 0.0;
    const size_t n = v.size() / 2;
    std::nth_element(v.begin(), v.begin() + n, v.end());
    double med = v[n];
    if (v.size() % 2 == 0) {
        std::nth_element(v.begin(), v.begin() + n - 1, v.end());
        med = 0.5 * (med + v[n - 1]);
    }
    return med;
}

// median over the frame of |T(p) - p|, T applied about the frame centre like the aligner's transforms
// (imgproc.cpp:401-411).  The per-frame figure is element size/2 of the sorted magnitudes, without the even-size
// averaging (eval_jitter.cpp:58-64).
inline double flow_median(const vs_transform& t, int w, int h, int lattice = 33) {
    std::vector<float> mag;
    mag.reserve((size_t)lattice * lattice);
    const double cx = 0.5 * w, cy = 0.5 * h;
    for (int j = 0; j < lattice; j++) {
        for (int i = 0; i < lattice; i++) {
            const double u = (i + 0.5) * w / lattice - cx, v = (j + 0.5) * h / lattice - cy;
            const double dx = t.A * u - t.B * v + t.TX, dy = t.B * u + t.A * v + t.TY;
            mag.push_back((float)std::sqrt(dx * dx + dy * dy));
        }
    }
    const size_t n = mag.size() / 2;
    std::nth_element(mag.begin(), mag.begin() + n, mag.end());
    return mag[n];
}

// transforms[i] relates frame i to frame i-1 (entry 0, the first frame, is skipped)
inline double jitter(const vs_transform* transforms, int n_frames, int w, int h) {
    std::vector<double> meds;
    for (int i = 1; i < n_frames; i++) meds.push_back(flow_median(transforms[i], w, h));
    return median(meds);
}

}  // namespace vsjit
