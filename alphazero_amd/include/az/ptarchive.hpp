// ptarchive.hpp -- reads the checkpoint the reference's trainer writes with
// `torch::save(net, path)` (util.h:108-139) without libtorch and without running
// anything from the file: the generator (generate.cpp:73 -> util.h:60-105) can then
// take `<prefix>_<ver>.pt` as it is, no converter beside it.
//
// The archive is a zip of stored (uncompressed) entries:
//   <name>/data.pkl     pickle (protocol 2) of the module tree: nested objects whose
//                       state is a dict  attribute -> submodule | tensor
//   <name>/data/<key>   raw little-endian storage of tensor <key>
// A tensor in the pickle is torch._utils._rebuild_tensor_v2(storage, offset, sizes,
// strides, ...) with storage = persistent id ('storage', torch.<T>Storage, key, device,
// numel).  The pickle is DATA here: a small stack machine builds a value tree from the
// twenty-odd opcodes such archives use and refuses everything else; GLOBAL names are kept
// as strings and never resolved, REDUCE only records (callable name, arguments).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace az {
namespace pt {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

// ------------------------------------------------------------------ zip (stored entries only)
struct ZipEntry { std::string name; uint64_t size = 0, data_off = 0; bool stored = true; };

class Zip {
public:
    explicit Zip(const std::string& path) : f_(path, std::ios::binary) {
        if (!f_) throw Error("cannot open " + path);
        f_.seekg(0, std::ios::end);
        const uint64_t fsz = (uint64_t)f_.tellg();
        // end of central directory: signature 0x06054b50 within the last 64 KB + 22 bytes
        const uint64_t tail = fsz < 65557 ? fsz : 65557;
        std::vector<unsigned char> t(tail);
        read(fsz - tail, t.data(), tail);
        int64_t e = -1;
        for (int64_t i = (int64_t)tail - 22; i >= 0; --i)
            if (t[i] == 0x50 && t[i + 1] == 0x4b && t[i + 2] == 0x05 && t[i + 3] == 0x06) { e = i; break; }
        if (e < 0) throw Error(path + ": not a zip archive (no end-of-central-directory record)");
        uint64_t n = u16(&t[e + 10]), cd_size = u32(&t[e + 12]), cd_off = u32(&t[e + 16]);
        if (n == 0xFFFF || cd_size == 0xFFFFFFFFu || cd_off == 0xFFFFFFFFu) {
            // zip64: locator (0x07064b50) 20 bytes in front of the record above -> zip64 end record (0x06064b50)
            if (e < 20 || u32(&t[e - 20]) != 0x07064b50u) throw Error(path + ": zip64 locator missing");
            const uint64_t z = u64(&t[e - 20 + 8]);
            unsigned char r[56];
            read(z, r, 56);
            if (u32(r) != 0x06064b50u) throw Error(path + ": zip64 end record missing");
            n = u64(r + 32); cd_size = u64(r + 40); cd_off = u64(r + 48);
        }
        std::vector<unsigned char> cd(cd_size);
        read(cd_off, cd.data(), cd_size);
        uint64_t p = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (p + 46 > cd_size || u32(&cd[p]) != 0x02014b50u) throw Error(path + ": corrupt central directory");
            const unsigned method = u16(&cd[p + 10]);
            uint64_t csz = u32(&cd[p + 20]), usz = u32(&cd[p + 24]), lho = u32(&cd[p + 42]);
            const unsigned nl = u16(&cd[p + 28]), xl = u16(&cd[p + 30]), cl = u16(&cd[p + 32]);
            ZipEntry ze;
            ze.name.assign((const char*)&cd[p + 46], nl);
            // zip64 extra field 0x0001: the 8-byte values of whichever fields above are all ones, in this order
            for (uint64_t x = p + 46 + nl; x + 4 <= p + 46 + nl + xl;) {
                const unsigned id = u16(&cd[x]), len = u16(&cd[x + 2]);
                if (id == 1) {
                    uint64_t q = x + 4;
                    if (usz == 0xFFFFFFFFu) { usz = u64(&cd[q]); q += 8; }
                    if (csz == 0xFFFFFFFFu) { csz = u64(&cd[q]); q += 8; }
                    if (lho == 0xFFFFFFFFu) { lho = u64(&cd[q]); q += 8; }
                }
                x += 4 + len;
            }
            ze.stored = method == 0 && csz == usz;      // (the archive's code/ entries are deflated; nothing here reads them)
            unsigned char lh[30];
            read(lho, lh, 30);
            if (u32(lh) != 0x04034b50u) throw Error(path + ": corrupt local header of " + ze.name);
            ze.size = usz;
            ze.data_off = lho + 30 + u16(lh + 26) + u16(lh + 28);
            if (ze.stored && ze.data_off + ze.size > fsz) throw Error(path + ": entry " + ze.name + " runs past the end of the file");
            entries_.push_back(ze);
            p += 46 + nl + xl + cl;
        }
    }
    const std::vector<ZipEntry>& entries() const { return entries_; }
    const ZipEntry* find(const std::string& name) const {
        for (auto& e : entries_) if (e.name == name) return &e;
        return nullptr;
    }
    std::vector<unsigned char> bytes(const ZipEntry& e) {
        if (!e.stored) throw Error("entry " + e.name + " is compressed (torch::save stores the pickle and the tensors)");
        std::vector<unsigned char> b(e.size);
        if (e.size) read(e.data_off, b.data(), e.size);
        return b;
    }

private:
    static unsigned u16(const unsigned char* p) { return p[0] | (p[1] << 8); }
    static uint32_t u32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
    static uint64_t u64(const unsigned char* p) { return (uint64_t)u32(p) | ((uint64_t)u32(p + 4) << 32); }
    void read(uint64_t off, unsigned char* dst, uint64_t n) {
        f_.clear();
        f_.seekg((std::streamoff)off);
        if (!f_.read((char*)dst, (std::streamsize)n)) throw Error("short read in archive");
    }
    std::ifstream f_;
    std::vector<ZipEntry> entries_;
};

// ------------------------------------------------------------------ pickle as data
struct Val;
typedef std::shared_ptr<Val> V;
struct Val {
    enum Kind { NONE, BOOL, INT, STR, GLOBAL, TUPLE, LIST, DICT, OBJECT, REDUCED, PERSID, MARK } kind = NONE;
    int64_t i = 0;
    std::string s;                              // STR, GLOBAL ("module name")
    std::vector<V> items;                       // TUPLE, LIST; REDUCED / OBJECT: [callable or class, args]; PERSID: [id]
    std::vector<std::pair<V, V>> dict;          // DICT; OBJECT: the state passed to BUILD
};

inline V unpickle(const std::vector<unsigned char>& d) {
    std::vector<V> st;
    std::map<uint32_t, V> memo;
    size_t p = 0;
    auto need = [&](size_t n) { if (p + n > d.size()) throw Error("pickle ends inside an opcode"); };
    auto mk = [](Val::Kind k) { V v = std::make_shared<Val>(); v->kind = k; return v; };
    auto pop = [&]() { if (st.empty()) throw Error("pickle stack underflow"); V v = st.back(); st.pop_back(); return v; };
    auto pop_mark = [&]() {                     // items above the topmost MARK, in order
        std::vector<V> out;
        for (;;) {
            V v = pop();
            if (v->kind == Val::MARK) break;
            out.push_back(v);
        }
        return std::vector<V>(out.rbegin(), out.rend());
    };
    auto le = [&](size_t n) { need(n); uint64_t v = 0; for (size_t k = 0; k < n; ++k) v |= (uint64_t)d[p + k] << (8 * k); p += n; return v; };
    auto line = [&]() { size_t e = p; while (e < d.size() && d[e] != '\n') ++e; if (e >= d.size()) throw Error("unterminated line in pickle");
                        std::string s((const char*)&d[p], e - p); p = e + 1; return s; };
    for (;;) {
        need(1);
        const unsigned char op = d[p++];
        switch (op) {
        case 0x80: le(1); break;                                                   // PROTO
        case '.': { V v = pop(); return v; }                                       // STOP
        case 'N': st.push_back(mk(Val::NONE)); break;
        case 0x88: case 0x89: { V v = mk(Val::BOOL); v->i = op == 0x88; st.push_back(v); break; }
        case 'K': { V v = mk(Val::INT); v->i = (int64_t)le(1); st.push_back(v); break; }       // BININT1
        case 'M': { V v = mk(Val::INT); v->i = (int64_t)le(2); st.push_back(v); break; }       // BININT2
        case 'J': { V v = mk(Val::INT); v->i = (int32_t)le(4); st.push_back(v); break; }       // BININT
        case 0x8a: {                                                               // LONG1
            const size_t n = (size_t)le(1);
            if (n > 8) throw Error("integer wider than 64 bits in pickle");
            need(n);
            int64_t x = 0;
            for (size_t k = 0; k < n; ++k) x |= (int64_t)d[p + k] << (8 * k);
            if (n && n < 8 && (d[p + n - 1] & 0x80)) x |= -((int64_t)1 << (8 * n));
            p += n;
            V v = mk(Val::INT); v->i = x; st.push_back(v); break;
        }
        case 'X': { const size_t n = (size_t)le(4); need(n); V v = mk(Val::STR); v->s.assign((const char*)&d[p], n); p += n; st.push_back(v); break; }
        case 0x8c: { const size_t n = (size_t)le(1); need(n); V v = mk(Val::STR); v->s.assign((const char*)&d[p], n); p += n; st.push_back(v); break; }
        case 'c': { V v = mk(Val::GLOBAL); v->s = line(); v->s += " " + line(); st.push_back(v); break; }
        case 'q': { if (st.empty()) throw Error("BINPUT on an empty stack"); memo[(uint32_t)le(1)] = st.back(); break; }
        case 'r': { if (st.empty()) throw Error("LONG_BINPUT on an empty stack"); memo[(uint32_t)le(4)] = st.back(); break; }
        case 'h': { auto it = memo.find((uint32_t)le(1)); if (it == memo.end()) throw Error("BINGET of an unset slot"); st.push_back(it->second); break; }
        case 'j': { auto it = memo.find((uint32_t)le(4)); if (it == memo.end()) throw Error("LONG_BINGET of an unset slot"); st.push_back(it->second); break; }
        case '(': st.push_back(mk(Val::MARK)); break;
        case ')': st.push_back(mk(Val::TUPLE)); break;
        case ']': st.push_back(mk(Val::LIST)); break;
        case '}': st.push_back(mk(Val::DICT)); break;
        case 't': { V v = mk(Val::TUPLE); v->items = pop_mark(); st.push_back(v); break; }
        case 0x85: case 0x86: case 0x87: {                                         // TUPLE1..3
            V v = mk(Val::TUPLE);
            v->items.resize(op - 0x84);
            for (int k = op - 0x85; k >= 0; --k) v->items[k] = pop();
            st.push_back(v); break;
        }
        case 'a': { V x = pop(); if (st.empty() || st.back()->kind != Val::LIST) throw Error("APPEND to a non-list"); st.back()->items.push_back(x); break; }
        case 'e': {
            auto xs = pop_mark();
            if (st.empty() || st.back()->kind != Val::LIST) throw Error("APPENDS to a non-list");
            for (auto& x : xs) st.back()->items.push_back(x);
            break;
        }
        case 's': {
            V val = pop(), key = pop();
            if (st.empty() || st.back()->kind != Val::DICT) throw Error("SETITEM on a non-dict");
            st.back()->dict.push_back({key, val});
            break;
        }
        case 'u': {
            auto xs = pop_mark();
            if (xs.size() % 2 || st.empty() || st.back()->kind != Val::DICT) throw Error("SETITEMS on a non-dict");
            for (size_t k = 0; k < xs.size(); k += 2) st.back()->dict.push_back({xs[k], xs[k + 1]});
            break;
        }
        case 0x81: { V args = pop(), cls = pop(); V v = mk(Val::OBJECT); v->items = {cls, args}; st.push_back(v); break; }      // NEWOBJ
        case 'R': { V args = pop(), fn = pop(); V v = mk(Val::REDUCED); v->items = {fn, args}; st.push_back(v); break; }
        case 'b': { V state = pop(); if (st.empty()) throw Error("BUILD on an empty stack");
                    if (state->kind == Val::DICT) st.back()->dict = state->dict; else st.back()->items.push_back(state); break; }
        case 'Q': { V id = pop(); V v = mk(Val::PERSID); v->items = {id}; st.push_back(v); break; }
        default: {
            char b[64];
            snprintf(b, sizeof b, "pickle opcode 0x%02x at byte %zu is not one a checkpoint uses", op, p - 1);
            throw Error(b);
        }
        }
    }
}

// ------------------------------------------------------------------ module tree -> named fp32 tensors
struct Tensor { std::vector<int64_t> shape; std::vector<float> v; };

class Archive {
public:
    explicit Archive(const std::string& path) : zip_(path) {
        for (auto& e : zip_.entries()) {
            const std::string tail = "/data.pkl";
            if (e.name.size() > tail.size() && e.name.compare(e.name.size() - tail.size(), tail.size(), tail) == 0) {
                prefix_ = e.name.substr(0, e.name.size() - tail.size());
                walk(unpickle(zip_.bytes(e)), "");
                return;
            }
        }
        throw Error(path + ": no data.pkl (not a torch::save archive of this libtorch generation)");
    }
    // parameters and buffers by dotted name, e.g. "RBlock_3.bn1.running_var"
    const std::map<std::string, Tensor>& tensors() const { return t_; }
    const Tensor& at(const std::string& name) const {
        auto it = t_.find(name);
        if (it == t_.end()) throw Error("checkpoint has no tensor " + name);
        return it->second;
    }
    bool has(const std::string& name) const { return t_.count(name) != 0; }

private:
    static bool is_global(const V& v, const char* name) { return v && v->kind == Val::GLOBAL && v->s == name; }
    void walk(const V& v, const std::string& name) {
        if (!v) return;
        if (v->kind == Val::OBJECT) {                                  // a module: attribute -> child
            for (auto& kv : v->dict)
                if (kv.first->kind == Val::STR) walk(kv.second, name.empty() ? kv.first->s : name + "." + kv.first->s);
            return;
        }
        if (v->kind == Val::REDUCED && v->items.size() == 2 && v->items[1]->kind == Val::TUPLE &&
            (is_global(v->items[0], "torch._utils _rebuild_tensor_v2") || is_global(v->items[0], "torch._utils _rebuild_tensor"))) {
            const auto& a = v->items[1]->items;
            if (a.size() < 4 || a[0]->kind != Val::PERSID) throw Error("tensor " + name + ": unexpected rebuild arguments");
            const V id = a[0]->items[0];
            if (id->kind != Val::TUPLE || id->items.size() < 5 || id->items[0]->kind != Val::STR || id->items[0]->s != "storage")
                throw Error("tensor " + name + ": unexpected storage reference");
            const std::string stype = id->items[1]->kind == Val::GLOBAL ? id->items[1]->s : "";
            const std::string key = id->items[2]->s;
            const int64_t numel = id->items[4]->i, off = a[1]->i;
            Tensor t;
            std::vector<int64_t> stride;
            for (auto& x : a[2]->items) t.shape.push_back(x->i);
            for (auto& x : a[3]->items) stride.push_back(x->i);
            if (stride.size() != t.shape.size()) throw Error("tensor " + name + ": sizes and strides disagree");
            size_t esz;
            enum { F32, F64, I64, I32 } ty;
            if (stype == "torch FloatStorage") { ty = F32; esz = 4; }
            else if (stype == "torch DoubleStorage") { ty = F64; esz = 8; }
            else if (stype == "torch LongStorage") { ty = I64; esz = 8; }
            else if (stype == "torch IntStorage") { ty = I32; esz = 4; }
            else throw Error("tensor " + name + ": storage type " + stype + " is not handled");
            const ZipEntry* e = zip_.find(prefix_ + "/data/" + key);
            if (!e) throw Error("tensor " + name + ": storage " + key + " is missing from the archive");
            if (e->size < (uint64_t)numel * esz) throw Error("tensor " + name + ": storage " + key + " is shorter than its pickle says");
            const std::vector<unsigned char> raw = zip_.bytes(*e);
            int64_t n = 1;
            for (int64_t s : t.shape) { if (s < 0) throw Error("tensor " + name + ": negative size"); n *= s; }
            t.v.resize((size_t)n);
            std::vector<int64_t> idx(t.shape.size(), 0);
            for (int64_t k = 0; k < n; ++k) {
                int64_t at = off;
                for (size_t dd = 0; dd < idx.size(); ++dd) at += idx[dd] * stride[dd];
                if (at < 0 || at >= numel) throw Error("tensor " + name + ": element outside its storage");
                const unsigned char* q = raw.data() + (size_t)at * esz;
                switch (ty) {
                case F32: { float x; memcpy(&x, q, 4); t.v[(size_t)k] = x; break; }
                case F64: { double x; memcpy(&x, q, 8); t.v[(size_t)k] = (float)x; break; }
                case I64: { int64_t x; memcpy(&x, q, 8); t.v[(size_t)k] = (float)x; break; }
                case I32: { int32_t x; memcpy(&x, q, 4); t.v[(size_t)k] = (float)x; break; }
                }
                for (int dd = (int)idx.size() - 1; dd >= 0; --dd) { if (++idx[dd] < t.shape[dd]) break; idx[dd] = 0; }
            }
            t_[name] = std::move(t);
            return;
        }
        // anything else (training flag, None hooks, ...) carries no weights
    }
    Zip zip_;
    std::string prefix_;
    std::map<std::string, Tensor> t_;
};

}  // namespace pt
}  // namespace az
