// Engine files (bs_zoedepth_forward / bs_cyclepose_forward for a non-Python host).  A plan -- the fixed launch
// sequence of one (model, batch, frame size, precision) over static device buffers -- is compiled once by the Python builder
// (bodyslam_amd/engine_export.py) and written down as buffers + launches; this file loads it and runs it through the library's own
// entry points.  No Python at inference time, no allocation per call: buffers are made at load, the two lanes and their fork / join
// events at load.  File layout: engine_export.py.
#include <stdio.h>
#include <string.h>

#include <exception>
#include <map>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <memory>
#include <vector>

#include "common.h"

namespace {

// (version 02, round 6: BS_ACT_SOFTPLUS became the exact form in round 5 and bs_gemm_desc grew out2_relu inside its tail padding -- a
// round-4 file would load and replay with other bits; the version is bumped whenever an enum's meaning or the descriptor's layout changes)
constexpr char kMagic[8] = {'B', 'S', 'E', 'N', 'G', '0', '2', '\0'};
enum { KIND_WORKSPACE = 0, KIND_ZERO = 1, KIND_DATA = 2 };
enum { ARG_I64 = 0, ARG_F64 = 1, ARG_PTR = 2, ARG_NULL = 3, ARG_DESC = 4 };
enum { OP_CALL = 0, OP_SIGNAL = 1, OP_WAIT = 2 };

struct Arg {
    int64_t i = 0;
    double d = 0.0;
    void* p = nullptr;
};

// ---- calling an entry point from a packed argument list: one trampoline per signature, generated from the prototype -----------------
template <typename T>
T arg_as(const Arg& a) {
    if constexpr (std::is_pointer<T>::value) return reinterpret_cast<T>(a.p);
    else if constexpr (std::is_floating_point<T>::value) return static_cast<T>(a.d);
    else return static_cast<T>(a.i);
}
template <typename... Args, size_t... I>
int call_unpacked(int (*fn)(Args...), const Arg* args, void* st, std::index_sequence<I...>) {
    using Tup = std::tuple<Args...>;           // Args = the prototype's parameters; the last one is the stream
    return fn(arg_as<std::tuple_element_t<I, Tup>>(args[I])..., st);
}
struct Entry {
    int nargs;
    int (*thunk)(const void* fn, const Arg* args, void* st);
    const void* fn;
};
template <typename... Args>
Entry make_entry(int (*fn)(Args...)) {
    static_assert(sizeof...(Args) >= 1, "an entry point takes the stream last");
    Entry e;
    e.nargs = (int)sizeof...(Args) - 1;
    e.fn = reinterpret_cast<const void*>(fn);
    e.thunk = [](const void* f, const Arg* args, void* st) {
        return call_unpacked<Args...>(reinterpret_cast<int (*)(Args...)>(const_cast<void*>(f)), args, st, std::make_index_sequence<sizeof...(Args) - 1>{});
    };
    return e;
}
#define BS_ENGINE_ENTRY(name) {#name, make_entry(name)}
const std::map<std::string, Entry>& entries() {
    // every entry point a plan of zoedepth.py / cyclepose.py issues (all take the stream last)
    static const std::map<std::string, Entry> m = {
        BS_ENGINE_ENTRY(bs_gemm),           BS_ENGINE_ENTRY(bs_attention),          BS_ENGINE_ENTRY(bs_attention_table),
        BS_ENGINE_ENTRY(bs_attention_table_corr),
        BS_ENGINE_ENTRY(bs_layernorm),      BS_ENGINE_ENTRY(bs_cast),               BS_ENGINE_ENTRY(bs_copy_f32),
        BS_ENGINE_ENTRY(bs_cast_split),     BS_ENGINE_ENTRY(bs_relu_split),         BS_ENGINE_ENTRY(bs_preprocess_patches),
        BS_ENGINE_ENTRY(bs_fill_rows),      BS_ENGINE_ENTRY(bs_upconv_tapsum),      BS_ENGINE_ENTRY(bs_small_attention),
        BS_ENGINE_ENTRY(bs_route_argmax),   BS_ENGINE_ENTRY(bs_resize_bilinear_nhwc), BS_ENGINE_ENTRY(bs_rank1_bias),
        BS_ENGINE_ENTRY(bs_postprocess_depth), BS_ENGINE_ENTRY(bs_logbinom_depth_ex), BS_ENGINE_ENTRY(bs_col_mean),
        BS_ENGINE_ENTRY(bs_attractor_step), BS_ENGINE_ENTRY(bs_add_resized),        BS_ENGINE_ENTRY(bs_instnorm_relu_nhwc),
        BS_ENGINE_ENTRY(bs_cyclepose_im2col), BS_ENGINE_ENTRY(bs_cyclepose_im2col_window), BS_ENGINE_ENTRY(bs_cyclepose_head),
        BS_ENGINE_ENTRY(bs_avgpool_nhwc),   BS_ENGINE_ENTRY(bs_mlp2),            BS_ENGINE_ENTRY(bs_mlp2_add),
        BS_ENGINE_ENTRY(bs_upconv_fused),   BS_ENGINE_ENTRY(bs_resize_bias_relu_nhwc), BS_ENGINE_ENTRY(bs_projector_level),
    };
    return m;
}

struct Op {
    int kind = OP_CALL, lane = 0;
    const Entry* entry = nullptr;
    std::vector<Arg> args;
    std::vector<char> desc;        // bs_gemm: the relocated descriptor (args[0].p points at it)
    std::string name;
};
struct Io {
    void* ptr;
    int64_t nbytes;
};

}  // namespace

struct bs_engine {
    int device = 0;
    std::vector<void*> bufs;
    std::vector<Op> ops;
    std::map<std::string, Io> io;
    hipStream_t side = nullptr;
    std::map<int, hipEvent_t> events;
    int64_t device_bytes = 0;
};

namespace {

struct Reader {
    FILE* f;
    bool ok = true;
    template <typename T>
    T get() {
        T v{};
        if (ok && fread(&v, sizeof(T), 1, f) != 1) ok = false;
        return v;
    }
    void bytes(void* dst, size_t n) {
        if (ok && n && fread(dst, 1, n, f) != n) ok = false;
    }
};

void engine_free(bs_engine* e) {
    if (!e) return;
    for (void* p : e->bufs)
        if (p) (void)hipFree(p);
    for (auto& kv : e->events) (void)hipEventDestroy(kv.second);
    if (e->side) (void)hipStreamDestroy(e->side);
    delete e;
}

}  // namespace

using namespace bs;

static int engine_load_impl(const char* path, bs_engine** out) {
    if (!initialized()) { set_error("bs_engine_load: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(path && out, "bs_engine_load: null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    BS_REQUIRE(f, "bs_engine_load: cannot open %s", path);
    // the file and the partly built engine are owned by guards: every early return AND an exception out of a std::vector / new (caught at
    // the C boundary below) closes the file and frees the device buffers (round-4 advisor)
    std::unique_ptr<FILE, int (*)(FILE*)> file_guard(f, fclose);
    Reader r{f};
    char magic[8];
    r.bytes(magic, 8);
    const uint32_t n_buf = r.get<uint32_t>(), n_ops = r.get<uint32_t>(), n_io = r.get<uint32_t>(), desc_size = r.get<uint32_t>();
    const int64_t calls_bytes = r.get<int64_t>();
    // the file is untrusted input of a public C entry point: every count is bounded by the file's size before anything is allocated
    long file_size = 0;
    {
        const long pos = ftell(f);
        if (fseek(f, 0, SEEK_END) == 0) file_size = ftell(f);
        (void)fseek(f, pos, SEEK_SET);
    }
    if (r.ok && memcmp(magic, kMagic, 8) == 0 &&
        ((long)n_buf * 20 > file_size || (long)n_io * 52 > file_size || (long)n_ops * 56 > file_size || calls_bytes < 0 || calls_bytes > file_size)) {
        set_error("bs_engine_load: %s: record counts exceed the file size (corrupt or truncated)", path);
        return BS_ERR_INVALID;
    }
    if (!r.ok || memcmp(magic, kMagic, 8) != 0) {
        set_error("bs_engine_load: %s is not an engine file of this version", path);
        return BS_ERR_INVALID;
    }
    if (desc_size != sizeof(bs_gemm_desc)) {
        set_error("bs_engine_load: %s was exported against a bs_gemm_desc of %u bytes, this library's has %zu: re-export it", path, desc_size,
                  sizeof(bs_gemm_desc));
        return BS_ERR_INVALID;
    }
    std::unique_ptr<bs_engine, void (*)(bs_engine*)> engine_guard(new bs_engine(), engine_free);
    bs_engine* e = engine_guard.get();
    auto fail = [&](const char* what) {
        set_error("bs_engine_load: %s (%s)", what, path);
        return BS_ERR_INVALID;
    };
    (void)hipGetDevice(&e->device);
    struct BufRec { int64_t nbytes; uint32_t kind; int64_t off; };
    std::vector<BufRec> recs(n_buf);
    for (auto& b : recs) {
        b.nbytes = r.get<int64_t>();
        b.kind = r.get<uint32_t>();
        b.off = r.get<int64_t>();
    }
    struct IoRec { char name[32]; uint32_t buf; int64_t off, nbytes; };
    std::vector<IoRec> ios(n_io);
    for (auto& i : ios) {
        r.bytes(i.name, 32);
        i.buf = r.get<uint32_t>();
        i.off = r.get<int64_t>();
        i.nbytes = r.get<int64_t>();
    }
    if (!r.ok) return fail("truncated header");
    for (auto& b : recs)
        if (b.nbytes < 0 || (b.kind == KIND_DATA && (b.off < 0 || b.off > file_size || b.nbytes > file_size - b.off))) return fail("bad buffer record");
    for (auto& i : ios)
        if (i.buf >= n_buf || i.nbytes < 0 || i.off < 0 || i.off > recs[i.buf].nbytes || i.nbytes > recs[i.buf].nbytes - i.off) return fail("io record outside its buffer");
    // buffers
    e->bufs.assign(n_buf, nullptr);
    for (uint32_t i = 0; i < n_buf; ++i) {
        const size_t n = recs[i].nbytes > 0 ? (size_t)recs[i].nbytes : 16;
        if (hipMalloc(&e->bufs[i], n) != hipSuccess) return fail("out of device memory");
        e->device_bytes += (int64_t)n;
        if (recs[i].kind != KIND_DATA && hipMemset(e->bufs[i], 0, n) != hipSuccess) return fail("hipMemset failed");
    }
    auto at = [&](uint32_t b, int64_t off) -> void* {
        return (b < n_buf && off >= 0 && off <= recs[b].nbytes) ? static_cast<char*>(e->bufs[b]) + off : nullptr;
    };
    for (auto& i : ios) {
        i.name[31] = 0;
        void* p = at(i.buf, i.off);
        if (!p) return fail("bad io record");
        e->io[i.name] = Io{p, i.nbytes};
    }
    // launches
    const auto& tab = entries();
    e->ops.resize(n_ops);
    for (auto& op : e->ops) {
        op.kind = (int)r.get<uint32_t>();
        op.lane = (int)r.get<uint32_t>();
        char name[48];
        r.bytes(name, 48);
        name[47] = 0;
        op.name = name;
        const uint32_t nargs = r.get<uint32_t>();
        if (!r.ok || nargs > 64) return fail("truncated launch list");
        op.args.resize(nargs);
        for (auto& a : op.args) {
            const uint32_t ty = r.get<uint32_t>();
            if (ty == ARG_I64) {
                a.i = r.get<int64_t>();
            } else if (ty == ARG_F64) {
                a.d = r.get<double>();
            } else if (ty == ARG_NULL) {
                (void)r.get<int64_t>();
            } else if (ty == ARG_PTR) {
                const uint32_t b = r.get<uint32_t>();
                const int64_t off = r.get<int64_t>();
                a.p = at(b, off);
                if (!a.p) return fail("pointer argument outside its buffer");
            } else if (ty == ARG_DESC) {
                const uint32_t n = r.get<uint32_t>();
                if (!r.ok || n != sizeof(bs_gemm_desc)) return fail("descriptor size");
                op.desc.resize(n);
                r.bytes(op.desc.data(), n);
                const uint32_t nrel = r.get<uint32_t>();
                if (!r.ok || nrel > 64) return fail("descriptor relocations");
                for (uint32_t k = 0; k < nrel; ++k) {
                    const uint32_t foff = r.get<uint32_t>(), b = r.get<uint32_t>();
                    const int64_t off = r.get<int64_t>();
                    void* p = at(b, off);
                    if (!p || foff + sizeof(void*) > n) return fail("descriptor pointer outside its buffer");
                    memcpy(op.desc.data() + foff, &p, sizeof(void*));
                }
            } else {
                return fail("unknown argument type");
            }
        }
        if (!r.ok) return fail("truncated launch list");
        if (op.kind == OP_CALL) {
            auto it = tab.find(op.name);
            if (it == tab.end()) {
                set_error("bs_engine_load: the engine calls %s, which this library's engine runner does not know", name);
                return BS_ERR_INVALID;
            }
            op.entry = &it->second;
            if ((int)op.args.size() != op.entry->nargs) return fail("argument count does not match the entry point");
        } else if (op.args.size() != 1) {
            return fail("bad event record");
        }
    }
    for (auto& op : e->ops)
        if (!op.desc.empty()) op.args[0].p = op.desc.data();       // (after the vector has stopped moving)
    // constants
    const long data0 = ((long)(8 + 4 * 4 + 8 + (long)n_buf * 20 + (long)n_io * 52 + calls_bytes) + 255) / 256 * 256;
    std::vector<char> host;
    for (uint32_t i = 0; i < n_buf; ++i) {
        if (recs[i].kind != KIND_DATA) continue;
        host.resize((size_t)recs[i].nbytes);
        if (fseek(f, data0 + (long)recs[i].off, SEEK_SET) != 0 || fread(host.data(), 1, host.size(), f) != host.size()) return fail("truncated constants");
        if (hipMemcpy(e->bufs[i], host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
    }
    file_guard.reset();
    // the zero-fills and uploads above ran on the null stream; bs_engine_run may be given a non-blocking stream that does not wait for it
    if (hipDeviceSynchronize() != hipSuccess) {
        set_error("bs_engine_load: device synchronisation failed");
        return BS_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess) {
        set_error("bs_engine_load: cannot create the side stream");
        return BS_ERR_HIP;
    }
    for (auto& op : e->ops)
        if (op.kind != OP_CALL && !e->events.count((int)op.args[0].i)) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                set_error("bs_engine_load: cannot create an event");
                return BS_ERR_HIP;
            }
            e->events[(int)op.args[0].i] = ev;
        }
    *out = engine_guard.release();
    return BS_OK;
}

extern "C" int bs_engine_load(const char* path, bs_engine** out) {
    try {          // (std::vector / new on a corrupt count must not unwind through the C boundary)
        return engine_load_impl(path, out);
    } catch (const std::exception& ex) {
        if (out) *out = nullptr;
        set_error("bs_engine_load: %s (%s)", ex.what(), path ? path : "");
        return BS_ERR_INVALID;
    }
}

extern "C" int bs_engine_destroy(bs_engine* e) {
    if (e) (void)hipDeviceSynchronize();
    engine_free(e);
    return BS_OK;
}

extern "C" int bs_engine_io(const bs_engine* e, const char* name, void** dev_ptr, int64_t* nbytes) {
    BS_REQUIRE(e && name, "bs_engine_io: null argument");
    auto it = e->io.find(name);
    BS_REQUIRE(it != e->io.end(), "bs_engine_io: the engine has no input / output called %s", name);
    if (dev_ptr) *dev_ptr = it->second.ptr;
    if (nbytes) *nbytes = it->second.nbytes;
    return BS_OK;
}

extern "C" int bs_engine_upload(bs_engine* e, const char* name, const void* host, int64_t nbytes) {
    BS_REQUIRE(e && name && host, "bs_engine_upload: null argument");
    auto it = e->io.find(name);
    BS_REQUIRE(it != e->io.end(), "bs_engine_upload: the engine has no input / output called %s", name);
    BS_REQUIRE(nbytes == it->second.nbytes, "bs_engine_upload: %s holds %lld bytes, got %lld", name, (long long)it->second.nbytes, (long long)nbytes);
    BS_CHECK_HIP(hipMemcpy(it->second.ptr, host, (size_t)nbytes, hipMemcpyHostToDevice));
    return BS_OK;
}

extern "C" int bs_engine_download(bs_engine* e, const char* name, void* host, int64_t nbytes) {
    BS_REQUIRE(e && name && host, "bs_engine_download: null argument");
    auto it = e->io.find(name);
    BS_REQUIRE(it != e->io.end(), "bs_engine_download: the engine has no input / output called %s", name);
    BS_REQUIRE(nbytes == it->second.nbytes, "bs_engine_download: %s holds %lld bytes, got %lld", name, (long long)it->second.nbytes, (long long)nbytes);
    BS_CHECK_HIP(hipDeviceSynchronize());
    BS_CHECK_HIP(hipMemcpy(host, it->second.ptr, (size_t)nbytes, hipMemcpyDeviceToHost));
    return BS_OK;
}

extern "C" int64_t bs_engine_device_bytes(const bs_engine* e) { return e ? e->device_bytes : 0; }

extern "C" int bs_engine_run(bs_engine* e, void* stream) {
    if (!initialized()) { set_error("bs_engine_run: call bs_init first"); return BS_ERR_NOT_INIT; }
    BS_REQUIRE(e, "bs_engine_run: null engine");
    int dev_now = -1;
    BS_CHECK_HIP(hipGetDevice(&dev_now));
    BS_REQUIRE(dev_now == e->device, "bs_engine_run: the engine was loaded on device %d, the current device is %d", e->device, dev_now);
    hipStream_t lanes[2] = {reinterpret_cast<hipStream_t>(stream), e->side};
    for (const Op& op : e->ops) {
        hipStream_t st = lanes[op.lane ? 1 : 0];
        if (op.kind == OP_CALL) {
            const int rc = op.entry->thunk(op.entry->fn, op.args.data(), st);
            if (rc != BS_OK) return rc;          // (the entry point has set the error text)
        } else if (op.kind == OP_SIGNAL) {
            BS_CHECK_HIP(hipEventRecord(e->events[(int)op.args[0].i], st));
        } else {
            BS_CHECK_HIP(hipStreamWaitEvent(st, e->events[(int)op.args[0].i], 0));
        }
    }
    return BS_OK;
}

// ---- the two calls SURVEY section 8(b) names: inputs and outputs are the caller's device buffers, copied to / from the engine's static ones on
// the caller's stream (asynchronously; a caller that wants no copy fills bs_engine_io("frames") itself and calls bs_engine_run) --------------
static int io_copy_in(bs_engine* e, const char* name, const void* src, int64_t nbytes, hipStream_t st, const char* who) {
    auto it = e->io.find(name);
    BS_REQUIRE(it != e->io.end(), "%s: this engine has no input called %s (wrong model?)", who, name);
    BS_REQUIRE(nbytes == it->second.nbytes, "%s: %s is %lld bytes, the engine was built for %lld (batch / frame size differ)", who, name, (long long)nbytes,
               (long long)it->second.nbytes);
    if (src != it->second.ptr) BS_CHECK_HIP(hipMemcpyAsync(it->second.ptr, src, (size_t)nbytes, hipMemcpyDeviceToDevice, st));
    return BS_OK;
}
static int io_copy_out(bs_engine* e, const char* name, void* dst, int64_t nbytes, hipStream_t st, const char* who) {
    auto it = e->io.find(name);
    BS_REQUIRE(it != e->io.end(), "%s: this engine has no output called %s (wrong model?)", who, name);
    BS_REQUIRE(nbytes == it->second.nbytes, "%s: %s is %lld bytes, the engine was built for %lld", who, name, (long long)nbytes, (long long)it->second.nbytes);
    if (dst != it->second.ptr) BS_CHECK_HIP(hipMemcpyAsync(dst, it->second.ptr, (size_t)nbytes, hipMemcpyDeviceToDevice, st));
    return BS_OK;
}

extern "C" int bs_zoedepth_forward(bs_engine* e, const uint8_t* frames_dev, int32_t B, int32_t H, int32_t W, float* depth_m_dev, uint16_t* depth_u16_dev,
                                   void* stream) {
    BS_REQUIRE(e && frames_dev && B > 0 && H > 0 && W > 0, "bs_zoedepth_forward: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t px = (int64_t)B * H * W;
    int rc = io_copy_in(e, "frames", frames_dev, px * 3, st, "bs_zoedepth_forward");
    if (rc != BS_OK) return rc;
    rc = bs_engine_run(e, stream);
    if (rc != BS_OK) return rc;
    if (depth_m_dev && (rc = io_copy_out(e, "depth_m", depth_m_dev, px * 4, st, "bs_zoedepth_forward")) != BS_OK) return rc;
    if (depth_u16_dev && (rc = io_copy_out(e, "depth_u16", depth_u16_dev, px * 2, st, "bs_zoedepth_forward")) != BS_OK) return rc;
    return BS_OK;
}

extern "C" int bs_cyclepose_forward(bs_engine* e, const uint8_t* frames_dev, int32_t n_frames, int32_t H, int32_t W, const int32_t* pairs_dev, int32_t P,
                                    float* T_rel_dev, void* stream) {
    BS_REQUIRE(e && frames_dev && pairs_dev && T_rel_dev && n_frames > 0 && H > 0 && W > 0 && P > 0, "bs_cyclepose_forward: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = io_copy_in(e, "frames", frames_dev, (int64_t)n_frames * H * W * 3, st, "bs_cyclepose_forward");
    if (rc != BS_OK) return rc;
    if ((rc = io_copy_in(e, "pairs", pairs_dev, (int64_t)P * 8, st, "bs_cyclepose_forward")) != BS_OK) return rc;
    if ((rc = bs_engine_run(e, stream)) != BS_OK) return rc;
    return io_copy_out(e, "T", T_rel_dev, (int64_t)P * 64, st, "bs_cyclepose_forward");
}
