// TEST INFRASTRUCTURE — CPU oracle. Never linked into or called by the product library.
//
// Restates src/pdf.rs:7-84 of the reference.
#pragma once
#include <memory>
#include "onb.hpp"
#include "ray.hpp"

namespace orc {

// pdf.rs:7-10
struct PDF {
    virtual ~PDF() = default;
    virtual float value(V3 direction) const = 0;
    virtual V3 generate() const = 0;
};

// pdf.rs:12-33 — value() claims cos/pi over the whole hemisphere although generate() only covers
// z >= sqrt(0.1), and returns 1e-4 (not 0) at or below the horizon.
struct CosinePDF : PDF {
    ONB uvw;
    explicit CosinePDF(V3 w) : uvw(ONB::build_from_w(w)) {}
    float value(V3 direction) const override {
        float cosine = dot(direction, uvw.w);
        return cosine <= 0.f ? 0.0001f : cosine / PI;
    }
    V3 generate() const override { return uvw.local_dir(random_cosine_direction()); }
};

// pdf.rs:35-57 — get_one() is None for every hittable that can appear as the lights list, so the
// pdf wraps the whole list.
struct HittablePDF : PDF {
    V3 o;
    SharedHittable obj;
    HittablePDF(V3 o_, SharedHittable obj_) : o(o_) {
        auto one = obj_->get_one();
        obj = one ? one : obj_;
    }
    float value(V3 direction) const override { return obj->pdf_val(o, direction); }
    V3 generate() const override { return obj->random(o); }
};

// pdf.rs:59-84
struct MixPDF : PDF {
    std::unique_ptr<PDF> pdfs[2];
    MixPDF(std::unique_ptr<PDF> p0, std::unique_ptr<PDF> p1) {
        pdfs[0] = std::move(p0);
        pdfs[1] = std::move(p1);
    }
    float value(V3 direction) const override {
        return 0.5f * pdfs[0]->value(direction) + 0.5f * pdfs[1]->value(direction);
    }
    V3 generate() const override { return get_rand() < 0.5f ? pdfs[0]->generate() : pdfs[1]->generate(); }
};

}  // namespace orc
