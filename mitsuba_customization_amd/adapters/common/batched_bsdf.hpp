// batched_bsdf.hpp — the wavefront extension a host renderer reaches with
// dynamic_cast<const BatchedBSDF *>(bsdf).  It is the batched entry into the plugin boundary
// (SURVEY.md §3.4): arrays are f32, xyzxyz… / uvuv…, host or device pointers (include/merl_hip.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>

class BatchedBSDF {
public:
    virtual ~BatchedBSDF() {}
    // BSDF::eval(bRec, ESolidAngle) for n pairs
    virtual void evalBatch(const float *wi, const float *wo, size_t n, float *rgb) const = 0;
    // BSDF::pdf(bRec, ESolidAngle) for n pairs
    virtual void pdfBatch(const float *wi, const float *wo, size_t n, float *pdf) const = 0;
    // Mitsuba 3's BSDF::eval_pdf for n pairs (one table lookup serves both)
    virtual void evalPdfBatch(const float *wi, const float *wo, size_t n, float *rgb, float *pdf) const = 0;
    // BSDF::sample(bRec, pdf, sample) for n pairs: writes wo, pdf and eval/pdf
    virtual void sampleBatch(const float *wi, const float *u, size_t n, float *wo, float *pdf, float *weight) const = 0;
    // the fused unit: eval + pdf of (wi, wo) and sample(wi, u)
    virtual void evalSampleBatch(const float *wi, const float *wo, const float *u, size_t n,
                                 float *rgb, float *pdf, float *wo2, float *pdf2, float *weight) const = 0;
    // the fused unit over a wavefront queue: slots queue[0 .. min(*count, capacity)) of slot-indexed DEVICE arrays;
    // count lives in device memory (include/merl_hip.h, mrl_eval_sample_queue)
    virtual void evalSampleQueue(const float *wi, const float *wo, const float *u,
                                 const uint32_t *queue, const uint32_t *count, size_t capacity,
                                 float *rgb, float *pdf, float *wo2, float *pdf2, float *weight) const = 0;
    // spectral materials (an RGL *_spec.bsdf file under a spectral variant): the fused unit at W wavelengths per unit,
    // wavelengths [n][W] in nm, values / weight [n][W] (include/merl_hip.h, mrl_eval_sample_spectral_batch)
    virtual void evalSampleSpectralBatch(const float *, const float *, const float *, const float *, int, size_t,
                                         float *, float *, float *, float *, float *) const
    {
        throw std::runtime_error("this BSDF holds no spectral material");
    }
    // device-pointer calls are asynchronous: wait for them
    virtual void synchronize() const = 0;
};
