// The plan object behind nf_plan_create (include/nemofilt.h), for the library's own translation units.
#pragma once
#include "nf_common.h"
#include "nf_fft.h"

#include <mutex>

struct nf_plan {
    int H, W;
    bool haveF, haveD;
    nf::Plan2D<float> pf;
    nf::Plan2D<double> pd;
    std::mutex mu;
};

// builds the line plans of one precision on first use
int nf_plan_ensure_prec(nf_plan* p, int prec);
