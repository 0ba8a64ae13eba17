// src/bsdfs/measured.cpp for Mitsuba 3: the RGL material database's adaptive-parameterisation BSDF (upstream's stock `measured`)
#include "measured_bsdf.hpp"

NAMESPACE_BEGIN(mitsuba)
MI_IMPLEMENT_CLASS_VARIANT(Measured, BSDF)
NAMESPACE_END(mitsuba)
MI_EXPORT_PLUGIN(Measured, "RGL measured BSDF, adaptive parameterisation (MI355X / libmerl_hip)")
