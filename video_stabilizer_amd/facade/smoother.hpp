// smoother.hpp -- drop-in for the reference's smoother.hpp:10-30 on the C ABI.
#pragma once
#include "imgproc.hpp"

class L1SmootherCenter {
public:
    L1SmootherCenter(int lagBehind, int lagAhead, double lambda = 1.0) : h_(vs_smoother_create(lagBehind, lagAhead, lambda)) {}
    ~L1SmootherCenter() { vs_smoother_destroy(h_); }
    L1SmootherCenter(const L1SmootherCenter&) = delete;
    L1SmootherCenter& operator=(const L1SmootherCenter&) = delete;
    // smoother.cpp:74-127: returns true when a finalized transform was produced
    bool update(const SimilarityTransform& meas, SimilarityTransform& outFinalized) {
        vs_transform out{0, 0, 0, 0};
        const int r = vs_smoother_update(h_, &meas.c(), &out);
        if (r == 1) outFinalized = SimilarityTransform::from(out);
        return r == 1;
    }
private:
    vs_smoother* h_;
};
