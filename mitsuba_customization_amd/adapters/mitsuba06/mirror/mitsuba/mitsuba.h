// ABI-MIRROR of the Mitsuba 0.6 declarations a BSDF plugin touches.
//
// This is NOT Mitsuba source: it is this repo's own minimal re-declaration of the public
// plugin-facing interface (class BSDF with virtual eval / sample x2 / pdf / configure,
// BSDFSamplingRecord, Spectrum, Frame, Properties, MTS_EXPORT_PLUGIN -> extern "C" CreateInstance /
// GetDescription), written from the public API so that adapters/mitsuba06/*.cpp compile and can be
// driven by tests in a container that has no Mitsuba tree (the reference's mitsuba/ gitlink is
// empty).  Against a real tree, build the same plugin sources with -DMERL_USE_REAL_MITSUBA and the
// tree's include path instead of this directory.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define MTS_NAMESPACE_BEGIN namespace mitsuba {
#define MTS_NAMESPACE_END }
#define MTS_EXPORT __attribute__((visibility("default")))

MTS_NAMESPACE_BEGIN

typedef float Float;

struct Vector {
    Float x, y, z;
    Vector() : x(0), y(0), z(0) {}
    Vector(Float x_, Float y_, Float z_) : x(x_), y(y_), z(z_) {}
    Float operator[](int i) const { return (&x)[i]; }
    Float &operator[](int i) { return (&x)[i]; }
};
typedef Vector Normal;
struct Point2 {
    Float x, y;
    Point2() : x(0), y(0) {}
    Point2(Float x_, Float y_) : x(x_), y(y_) {}
};

// RGB build (SPECTRUM_SAMPLES == 3): linear RGB coefficients
class Spectrum {
public:
    Spectrum() { s[0] = s[1] = s[2] = 0; }
    explicit Spectrum(Float v) { s[0] = s[1] = s[2] = v; }
    void fromLinearRGB(Float r, Float g, Float b) { s[0] = r; s[1] = g; s[2] = b; }
    void toLinearRGB(Float &r, Float &g, Float &b) const { r = s[0]; g = s[1]; b = s[2]; }
    Float operator[](int i) const { return s[i]; }
    Float &operator[](int i) { return s[i]; }
    Spectrum operator*(Float f) const { Spectrum r; for (int i = 0; i < 3; ++i) r.s[i] = s[i] * f; return r; }
    Spectrum operator/(Float f) const { Spectrum r; for (int i = 0; i < 3; ++i) r.s[i] = s[i] / f; return r; }
    bool isZero() const { return s[0] == 0 && s[1] == 0 && s[2] == 0; }
private:
    Float s[3];
};

struct Frame {
    static Float cosTheta(const Vector &v) { return v.z; }
};

struct Intersection {
    Vector wi;      // incident direction in the local shading frame
};

enum EMeasure { EInvalidMeasure = 0, ESolidAngle = 1, ELength = 2, EArea = 3, EDiscrete = 4 };
enum ETransportMode { ERadiance = 0, EImportance = 1 };

class Sampler;
class InstanceManager;

// ---- FileResolver / Thread: how a 0.6 plugin turns the scene's "filename" into a path
// (Thread::getThread()->getFileResolver()->resolve(name): the scene's directory and the data directories are
// searched; an absolute or unresolvable name comes back unchanged) ----
struct ResolvedPath {
    std::string m_path;
    const std::string &string() const { return m_path; }
};
class FileResolver {
public:
    // The real resolver is filled by the scene loader (the scene file's directory) and lives in the host's core
    // library.  The mirror has no such library — every plugin .so carries its own copy — so its search path comes
    // from the environment: MITSUBA_MIRROR_DATA_PATH = dir[:dir...] (what the test driver sets for "the scene's directory").
    FileResolver()
    {
        if (const char *env = std::getenv("MITSUBA_MIRROR_DATA_PATH")) {
            std::string all(env);
            size_t a = 0;
            while (a <= all.size()) {
                const size_t b = all.find(':', a);
                const std::string dir = all.substr(a, b == std::string::npos ? std::string::npos : b - a);
                if (!dir.empty()) m_dirs.push_back(dir);
                if (b == std::string::npos) break;
                a = b + 1;
            }
        }
    }
    void appendPath(const std::string &dir) { m_dirs.push_back(dir); }
    void prependPath(const std::string &dir) { m_dirs.insert(m_dirs.begin(), dir); }
    ResolvedPath resolve(const std::string &name) const
    {
        if (!name.empty() && name[0] != '/')
            for (const std::string &d : m_dirs) {
                const std::string candidate = d + "/" + name;
                if (FILE *f = std::fopen(candidate.c_str(), "rb")) { std::fclose(f); return { candidate }; }
            }
        return { name };
    }
private:
    std::vector<std::string> m_dirs;
};
class Thread {
public:
    static Thread *getThread() { static Thread t; return &t; }
    FileResolver *getFileResolver() { return &m_resolver; }
private:
    FileResolver m_resolver;
};

// ---- Stream: what serialize() writes to and the unserialising constructor reads from (network rendering
// ships scene objects to worker nodes this way).  The mirror's stream is an in-memory byte queue. ----
class Stream {
public:
    virtual ~Stream() {}
    void writeString(const std::string &v) { writeUInt((unsigned int)v.size()); m_bytes.insert(m_bytes.end(), v.begin(), v.end()); }
    std::string readString()
    {
        const unsigned int n = readUInt();
        need(n);
        std::string v(m_bytes.begin() + (std::ptrdiff_t)m_pos, m_bytes.begin() + (std::ptrdiff_t)(m_pos + n));
        m_pos += n;
        return v;
    }
    void writeInt(int v) { put(&v, sizeof v); }
    int readInt() { int v; get(&v, sizeof v); return v; }
    void writeUInt(unsigned int v) { put(&v, sizeof v); }
    unsigned int readUInt() { unsigned int v; get(&v, sizeof v); return v; }
    void writeFloat(Float v) { put(&v, sizeof v); }
    Float readFloat() { Float v; get(&v, sizeof v); return v; }
    size_t getSize() const { return m_bytes.size(); }
    size_t getPos() const { return m_pos; }
private:
    void put(const void *p, size_t n) { const char *c = static_cast<const char *>(p); m_bytes.insert(m_bytes.end(), c, c + n); }
    void need(size_t n) const { if (m_pos + n > m_bytes.size()) throw std::runtime_error("Stream: read past the end"); }
    void get(void *p, size_t n) { need(n); std::copy(m_bytes.begin() + (std::ptrdiff_t)m_pos, m_bytes.begin() + (std::ptrdiff_t)(m_pos + n), static_cast<char *>(p)); m_pos += n; }
    std::vector<char> m_bytes;
    size_t m_pos = 0;
};

// ---- Properties: the key/value bag the scene XML hands to a plugin constructor ----
class Properties {
public:
    Properties() {}
    explicit Properties(const std::string &pluginName) : m_plugin(pluginName) {}
    const std::string &getPluginName() const { return m_plugin; }
    bool hasProperty(const std::string &n) const { return m_str.count(n) || m_num.count(n); }
    void setString(const std::string &n, const std::string &v) { m_str[n] = v; }
    void setFloat(const std::string &n, Float v) { m_num[n] = v; }
    void setInteger(const std::string &n, int v) { m_num[n] = (double)v; }
    void setBoolean(const std::string &n, bool v) { m_num[n] = v ? 1.0 : 0.0; }
    std::string getString(const std::string &n) const
    {
        auto it = m_str.find(n);
        if (it == m_str.end()) throw std::runtime_error("Property \"" + n + "\" has not been specified!");
        return it->second;
    }
    std::string getString(const std::string &n, const std::string &def) const { auto it = m_str.find(n); return it == m_str.end() ? def : it->second; }
    Float getFloat(const std::string &n, Float def) const { auto it = m_num.find(n); return it == m_num.end() ? def : (Float)it->second; }
    int getInteger(const std::string &n, int def) const { auto it = m_num.find(n); return it == m_num.end() ? def : (int)it->second; }
    bool getBoolean(const std::string &n, bool def) const { auto it = m_num.find(n); return it == m_num.end() ? def : it->second != 0.0; }
private:
    std::string m_plugin;
    std::map<std::string, std::string> m_str;
    std::map<std::string, double> m_num;
};

// ---- Object / ConfigurableObject: reference counted plugin instances ----
class Object {
public:
    Object() : m_refCount(0) {}
    virtual ~Object() {}
    void incRef() const { ++m_refCount; }
    void decRef() const { if (--m_refCount <= 0) delete this; }
    int getRefCount() const { return m_refCount; }
    virtual std::string toString() const { return "Object[]"; }
private:
    mutable int m_refCount;
};

class ConfigurableObject : public Object {
public:
    explicit ConfigurableObject(const Properties &props) : m_properties(props) {}
    ConfigurableObject(Stream *, InstanceManager *) {}       // unserialising constructor
    virtual void configure() {}
    virtual void serialize(Stream *, InstanceManager *) const {}
    const Properties &getProperties() const { return m_properties; }
protected:
    Properties m_properties;
};

class BSDF;

struct BSDFSamplingRecord {
    const Intersection &its;
    Sampler *sampler;
    Vector wi, wo;
    Float eta;
    ETransportMode mode;
    unsigned int typeMask;
    int component;
    unsigned int sampledType;
    int sampledComponent;

    // sampling constructor: wo is produced by BSDF::sample
    explicit BSDFSamplingRecord(const Intersection &its_, Sampler *sampler_ = nullptr, ETransportMode mode_ = ERadiance)
        : its(its_), sampler(sampler_), wi(its_.wi), wo(), eta(1.0f), mode(mode_), typeMask(0xFFFFFFFFu), component(-1),
          sampledType(0), sampledComponent(-1) {}
    // query constructors: eval / pdf
    BSDFSamplingRecord(const Intersection &its_, const Vector &wo_, ETransportMode mode_ = ERadiance)
        : its(its_), sampler(nullptr), wi(its_.wi), wo(wo_), eta(1.0f), mode(mode_), typeMask(0xFFFFFFFFu), component(-1),
          sampledType(0), sampledComponent(-1) {}
    BSDFSamplingRecord(const Intersection &its_, const Vector &wi_, const Vector &wo_, ETransportMode mode_ = ERadiance)
        : its(its_), sampler(nullptr), wi(wi_), wo(wo_), eta(1.0f), mode(mode_), typeMask(0xFFFFFFFFu), component(-1),
          sampledType(0), sampledComponent(-1) {}
};

class BSDF : public ConfigurableObject {
public:
    enum EBSDFType {
        ENull = 0x00001, EDiffuseReflection = 0x00002, EDiffuseTransmission = 0x00004,
        EGlossyReflection = 0x00008, EGlossyTransmission = 0x00010, EDeltaReflection = 0x00020,
        EDeltaTransmission = 0x00040, EDelta1DReflection = 0x00080, EDelta1DTransmission = 0x00100,
        EAnisotropic = 0x01000, ESpatiallyVarying = 0x02000, ENonSymmetric = 0x04000,
        EFrontSide = 0x08000, EBackSide = 0x10000, EUsesSampler = 0x20000
    };
    enum ETypeCombinations {
        EReflection = EDiffuseReflection | EDeltaReflection | EDelta1DReflection | EGlossyReflection,
        ESmooth = EDiffuseReflection | EDiffuseTransmission | EGlossyReflection | EGlossyTransmission,
        EAll = 0xFFFFFFFF
    };

    explicit BSDF(const Properties &props)
        : ConfigurableObject(props), m_combinedType(0), m_usesRayDifferentials(false), m_ensureEnergyConservation(true) {}
    // unserialising constructor + its counterpart, as in Mitsuba 0.6's BSDF: one flag travels
    BSDF(Stream *stream, InstanceManager *manager)
        : ConfigurableObject(stream, manager), m_combinedType(0), m_usesRayDifferentials(false), m_ensureEnergyConservation(stream->readInt() != 0) {}
    void serialize(Stream *stream, InstanceManager *manager) const override
    {
        ConfigurableObject::serialize(stream, manager);
        stream->writeInt(m_ensureEnergyConservation ? 1 : 0);
    }

    virtual Spectrum eval(const BSDFSamplingRecord &bRec, EMeasure measure = ESolidAngle) const = 0;
    virtual Spectrum sample(BSDFSamplingRecord &bRec, const Point2 &sample) const = 0;
    virtual Spectrum sample(BSDFSamplingRecord &bRec, Float &pdf, const Point2 &sample) const = 0;
    virtual Float pdf(const BSDFSamplingRecord &bRec, EMeasure measure = ESolidAngle) const = 0;

    void configure() override
    {
        m_combinedType = 0;
        for (unsigned int c : m_components) m_combinedType |= c;
    }
    int getComponentCount() const { return (int)m_components.size(); }
    unsigned int getType() const { return m_combinedType; }
    unsigned int getType(int i) const { return m_components[(size_t)i]; }
    bool usesRayDifferentials() const { return m_usesRayDifferentials; }

protected:
    std::vector<unsigned int> m_components;
    unsigned int m_combinedType;
    bool m_usesRayDifferentials;
    bool m_ensureEnergyConservation;
};

MTS_NAMESPACE_END

// The entry points PluginManager looks up with dlsym() after dlopen()ing plugins/<name>.so
#define MTS_EXPORT_PLUGIN(name, descr)                                                              \
    extern "C" {                                                                                    \
    void MTS_EXPORT *CreateInstance(const mitsuba::Properties &props) { return new mitsuba::name(props); } \
    const char MTS_EXPORT *GetDescription() { return descr; }                                       \
    }
#define MTS_DECLARE_CLASS()
// Mitsuba 0.6 registers an unserialisation function per serialisable class (Class::unserialize ->
// new name(stream, manager)).  The mirror has no class registry; it exports the same function under a fixed
// symbol so a test host can rebuild a plugin instance from a stream the way a render worker would.
#define MTS_IMPLEMENT_CLASS_S(name, abstract, super)                                                \
    extern "C" MTS_EXPORT void *UnserializeInstance(mitsuba::Stream *stream, mitsuba::InstanceManager *manager) \
    {                                                                                               \
        return new mitsuba::name(stream, manager);                                                  \
    }
