// daachorse_amd.hpp — header-only C++17 façade over the C ABI (include/daachorse_amd.h).
//
// The reference is a Rust crate; no Rust toolchain exists in this image, so the host side that a
// crate user would touch is mirrored here in C++ with the crate's names, argument meaning and
// error behaviour (reference src/bytewise.rs, src/bytewise/builder.rs, src/bytewise/iter.rs and their
// src/charwise* twins):
//
//     auto pma = daachorse::DoubleArrayAhoCorasick::new_({"bcd", "ab", "a"}).value();
//     for (auto it = pma.find_overlapping_iter("abcd"); auto m = it.next();)
//         use(m->start(), m->end(), m->value());
//
//   * construction / deserialisation return Result<T> (a value or a DaachorseError), like the
//     crate's `Result<_, DaachorseError>` (src/errors.rs:10-22, 140);
//   * calling a query of the wrong MatchKind PANICS in the crate (bytewise.rs:194-197, 299-302,
//     551-554): here it throws daachorse::PanicError carrying the crate's message;
//   * iterators are lazy (`next()`), borrow the automaton, and keep the haystack alive.
// Every scan runs on the MI355X; there is no CPU scan path behind this header.
#pragma once

#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <variant>
#include <vector>

#include "daachorse_amd.h"

namespace daachorse {

enum class MatchKind : uint8_t { Standard = 0, LeftmostLongest = 1, LeftmostFirst = 2 };  // src/lib.rs:324-346

struct DaachorseError {  // src/errors.rs:10-22
    enum Kind { InvalidArgument = 1, AutomatonScale = 2, InvalidConversion = 3, InvalidAutomaton = 4, Unsupported = 6, Device = 7 } kind;
    std::string message;
};

struct PanicError : std::runtime_error {  // the crate's assert!/panic! paths
    using std::runtime_error::runtime_error;
};

template <class T>
class Result {
public:
    Result(T v) : v_(std::move(v)) {}
    Result(DaachorseError e) : v_(std::move(e)) {}
    bool is_ok() const { return v_.index() == 0; }
    bool is_err() const { return !is_ok(); }
    T &value() {
        if (is_err()) throw PanicError("called `Result::unwrap()` on an `Err` value: " + std::get<1>(v_).message);
        return std::get<0>(v_);
    }
    T unwrap() { return std::move(value()); }
    const DaachorseError &error() const { return std::get<1>(v_); }

private:
    std::variant<T, DaachorseError> v_;
};

class Match {  // src/lib.rs:286-320
public:
    Match(uint64_t s, uint64_t e, uint32_t v) : s_(s), e_(e), v_(v) {}
    uint64_t start() const { return s_; }
    uint64_t end() const { return e_; }
    uint32_t value() const { return v_; }
    bool operator==(const Match &o) const { return s_ == o.s_ && e_ == o.e_ && v_ == o.v_; }

private:
    uint64_t s_, e_;
    uint32_t v_;
};

namespace detail {
inline DaachorseError make_error(daac_status st) {
    const char *m = daac_last_error();
    return DaachorseError{static_cast<DaachorseError::Kind>(st), m ? m : ""};
}
struct PmaDeleter { void operator()(daac_pma *p) const { daac_pma_free(p); } };
struct IterDeleter { void operator()(daac_iter *p) const { daac_iter_close(p); } };
}  // namespace detail

// Iterator<Item = Match<u32>>: FindIterator / FindOverlappingIterator / FindOverlappingNoSuffixIterator /
// LeftmostFindIterator of src/bytewise/iter.rs, one type here because the device engine sits behind them.
class MatchIterator {
public:
    // one 8-byte read from the run at hand (the compact iterator: {value, end relative to the run's base | length << end_bits}: half the
    // bytes over PCIe); a library call per window of the haystack (daac_iter_next_batch8), not per match.  Dictionaries with patterns of
    // several KB have no compact form: 16-byte runs then.
    std::optional<Match> next() {
        if (run_n_ == 0) {
            const int r = compact_ ? daac_iter_next_batch8(it_.get(), &run8_, &run_n_, &base_, &end_bits_) : daac_iter_next_batch(it_.get(), &run16_, &run_n_);
            if (r == 0) return std::nullopt;
            if (r < 0) throw PanicError(std::string("device scan failed: ") + daac_last_error());
        }
        --run_n_;
        if (!compact_) {
            const daac_match16 t = *run16_++;
            return Match(t.end - t.length, t.end, t.value);
        }
        const daac_match8 t = *run8_++;
        const uint64_t end = base_ + (t.end_len & ((1u << end_bits_) - 1u));
        return Match(end - (t.end_len >> end_bits_), end, t.value);
    }
    std::vector<Match> collect() {
        std::vector<Match> out;
        while (auto m = next()) out.push_back(*m);
        return out;
    }

private:
    template <class F>
    friend class BasicAhoCorasick;
    MatchIterator(daac_iter *it, std::unique_ptr<std::string> hay, bool compact) : hay_(std::move(hay)), it_(it), compact_(compact) {}
    std::unique_ptr<std::string> hay_;  // the haystack lives (at a stable address) as long as the iterator (the crate's `P`)
    std::unique_ptr<daac_iter, detail::IterDeleter> it_;
    bool compact_ = true;
    const daac_match8 *run8_ = nullptr;    // what is left of the last run (a view of the iterator's window buffer)
    const daac_match16 *run16_ = nullptr;
    size_t run_n_ = 0;
    uint64_t base_ = 0;
    uint32_t end_bits_ = 32;
};

// FindStepper / FindOverlappingStepper (src/bytewise/iter.rs:344-475, src/charwise/iter.rs:403-534), fed a chunk
// at a time: feed() returns the matches decided so far, positions counted from the first byte ever fed.
class Stepper {
public:
    std::vector<Match> feed(std::string_view chunk) {
        daac_matches *m = nullptr;
        const daac_status st = daac_stream_feed(s_.get(), reinterpret_cast<const uint8_t *>(chunk.data()), chunk.size(), 0, &m);
        if (st != DAAC_OK) throw PanicError(std::string("device scan failed: ") + daac_last_error());
        std::vector<Match> out;
        const daac_match *p = daac_matches_data(m);
        for (size_t i = 0, n = daac_matches_count(m); i < n; ++i) out.emplace_back(p[i].start, p[i].end, p[i].value);
        daac_matches_free(m);
        return out;
    }

private:
    template <class F>
    friend class BasicAhoCorasick;
    struct Closer { void operator()(daac_stream *s) const { daac_stream_close(s); } };
    explicit Stepper(daac_stream *s) : s_(s) {}
    std::unique_ptr<daac_stream, Closer> s_;
};

namespace detail {
// Which pair of C entry points builds / parses the automaton behind a handle (the scans are shared).
struct Bytewise {  // src/bytewise.rs, src/bytewise/builder.rs
    static daac_status parse(const uint8_t *b, size_t n, daac_pma **h, size_t *used) { return daac_bytewise_from_serialized(b, n, h, used); }
    static daac_status build(const uint8_t *b, const uint64_t *o, const uint32_t *v, size_t n, uint8_t k, uint32_t f, daac_pma **h) {
        return daac_bytewise_build(b, o, v, n, k, f, h);
    }
};
struct Charwise {  // src/charwise.rs, src/charwise/builder.rs
    static daac_status parse(const uint8_t *b, size_t n, daac_pma **h, size_t *used) { return daac_charwise_from_serialized(b, n, h, used); }
    static daac_status build(const uint8_t *b, const uint64_t *o, const uint32_t *v, size_t n, uint8_t k, uint32_t f, daac_pma **h) {
        return daac_charwise_build(b, o, v, n, k, f, h);
    }
};
}  // namespace detail

template <class Flavor>
class BasicBuilder;

// DoubleArrayAhoCorasick<u32> (src/bytewise.rs:54-68) and CharwiseDoubleArrayAhoCorasick<u32>
// (src/charwise.rs:59-65): the same surface; charwise haystacks and patterns are UTF-8, positions are bytes.
template <class Flavor>
class BasicAhoCorasick {
public:
    // bytewise.rs:103-110 / 154-161, charwise.rs:87-94 / 139-146 (`new` is a keyword in C++)
    static Result<BasicAhoCorasick> new_(const std::vector<std::string> &patterns) { return BasicBuilder<Flavor>().build(patterns); }
    static Result<BasicAhoCorasick> with_values(const std::vector<std::pair<std::string, uint32_t>> &patvals) {
        return BasicBuilder<Flavor>().build_with_values(patvals);
    }

    // bytewise.rs:868-964, charwise.rs:896-952: (automaton, bytes consumed)
    static Result<std::pair<BasicAhoCorasick, size_t>> deserialize(std::string_view source) {
        daac_pma *h = nullptr;
        size_t consumed = 0;
        const daac_status st = Flavor::parse(reinterpret_cast<const uint8_t *>(source.data()), source.size(), &h, &consumed);
        if (st != DAAC_OK) return detail::make_error(st);
        return std::make_pair(BasicAhoCorasick(h), consumed);
    }
    std::string serialize() const {  // bytewise.rs:801-820, charwise.rs:831-848
        uint8_t *buf = nullptr;
        size_t len = 0;
        if (daac_pma_serialize(h_.get(), &buf, &len) != DAAC_OK) throw PanicError(daac_last_error());
        std::string out(reinterpret_cast<char *>(buf), len);
        daac_free(buf);
        return out;
    }

    MatchIterator find_iter(std::string haystack) const { return open(DAAC_FIND, std::move(haystack), "Error: match_kind must be standard."); }
    MatchIterator find_overlapping_iter(std::string haystack) const {
        return open(DAAC_FIND_OVERLAPPING, std::move(haystack), "Error: match_kind must be standard.");
    }
    MatchIterator find_overlapping_no_suffix_iter(std::string haystack) const {
        return open(DAAC_FIND_OVERLAPPING_NO_SUFFIX, std::move(haystack), "Error: match_kind must be standard.");
    }
    MatchIterator leftmost_find_iter(std::string haystack) const {
        return open(DAAC_LEFTMOST_FIND, std::move(haystack), "Error: match_kind must be leftmost.");
    }

    // `pma.find_overlapping_iter(h).count()` and its three siblings without materialising a match (Iterator::count on the
    // iterators of src/bytewise/iter.rs / src/charwise/iter.rs): one device pass, nothing but the number comes back.
    size_t find_iter_count(std::string_view haystack) const { return count(DAAC_FIND, haystack, "Error: match_kind must be standard."); }
    size_t find_overlapping_iter_count(std::string_view haystack) const {
        return count(DAAC_FIND_OVERLAPPING, haystack, "Error: match_kind must be standard.");
    }
    size_t find_overlapping_no_suffix_iter_count(std::string_view haystack) const {
        return count(DAAC_FIND_OVERLAPPING_NO_SUFFIX, haystack, "Error: match_kind must be standard.");
    }
    size_t leftmost_find_iter_count(std::string_view haystack) const {
        return count(DAAC_LEFTMOST_FIND, haystack, "Error: match_kind must be leftmost.");
    }

    // The whole match list of find_overlapping_iter, in the iterator's order, left in device memory for a consumer that
    // runs on the GPU (daac_scan_device); `haystack_dev` is a device pointer.  to_host() copies a slice back.
    class DeviceMatches {
    public:
        DeviceMatches(DeviceMatches &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
        DeviceMatches(const DeviceMatches &) = delete;
        ~DeviceMatches() { daac_device_free(p_); }
        const daac_match *data() const { return p_; }  // device pointer
        size_t size() const { return n_; }
        std::vector<Match> to_host(size_t first, size_t n) const {
            if (first > n_ || n > n_ - first) throw std::out_of_range("DeviceMatches::to_host");
            std::vector<daac_match> raw(n);
            if (n && daac_device_to_host(raw.data(), p_ + first, n * sizeof(daac_match)) != DAAC_OK) throw PanicError(daac_last_error());
            std::vector<Match> out;
            out.reserve(n);
            for (const auto &m : raw) out.emplace_back(m.start, m.end, m.value);
            return out;
        }

    private:
        friend class BasicAhoCorasick;
        DeviceMatches(daac_match *p, size_t n) : p_(p), n_(n) {}
        daac_match *p_;
        size_t n_;
    };
    DeviceMatches find_overlapping_device(const uint8_t *haystack_dev, size_t len, void *stream = nullptr) const {
        return scan_device(haystack_dev, len, 1, stream);
    }
    DeviceMatches find_overlapping_device(std::string_view host_haystack) const {  // the haystack is copied over first
        return scan_device(reinterpret_cast<const uint8_t *>(host_haystack.data()), host_haystack.size(), 0, nullptr);
    }

    // bytewise.rs:238-251, 353-375 / charwise.rs: steppers for text that arrives in pieces
    Stepper find_stepper() const { return open_stepper(DAAC_FIND); }
    Stepper find_overlapping_stepper() const { return open_stepper(DAAC_FIND_OVERLAPPING); }

    MatchKind match_kind() const { return static_cast<MatchKind>(info().match_kind); }
    size_t num_states() const { return info().num_states; }
    size_t heap_bytes() const { return info().heap_bytes; }
    daac_pma *raw() const { return h_.get(); }

private:
    friend class BasicBuilder<Flavor>;
    explicit BasicAhoCorasick(daac_pma *h) : h_(h) {}
    daac_info info() const {
        daac_info i;
        i.struct_size = static_cast<uint32_t>(sizeof(i));
        daac_pma_info(h_.get(), &i);
        return i;
    }
    size_t count(int mode, std::string_view hay, const char *panic_msg) const {
        uint64_t n = 0;
        const daac_status st = daac_scan_count_only_range(h_.get(), mode, DAAC_ENGINE_AUTO, reinterpret_cast<const uint8_t *>(hay.data()),
                                                          hay.size(), 0, 0, nullptr, &n, nullptr);
        if (st == DAAC_ERR_MATCH_KIND) throw PanicError(panic_msg);
        if (st != DAAC_OK) throw PanicError(std::string("device scan failed: ") + daac_last_error());
        return n;
    }
    DeviceMatches scan_device(const uint8_t *hay, size_t len, int on_device, void *stream) const {
        daac_match *p = nullptr;
        uint64_t n = 0;
        const daac_status st = daac_scan_device(h_.get(), DAAC_FIND_OVERLAPPING, DAAC_ENGINE_AUTO, hay, len, on_device, stream, &p, &n);
        if (st == DAAC_ERR_MATCH_KIND) throw PanicError("Error: match_kind must be standard.");
        if (st != DAAC_OK) throw PanicError(std::string("device scan failed: ") + daac_last_error());
        return DeviceMatches(p, n);
    }
    Stepper open_stepper(int mode) const {
        daac_stream *st = nullptr;
        const daac_status rc = daac_stream_open(h_.get(), mode, DAAC_ENGINE_AUTO, nullptr, &st);
        if (rc == DAAC_ERR_MATCH_KIND) throw PanicError("Error: match_kind must be standard.");
        if (rc != DAAC_OK) throw PanicError(std::string("device scan failed: ") + daac_last_error());
        return Stepper(st);
    }
    MatchIterator open(int mode, std::string hay, const char *panic_msg) const {
        auto keep = std::make_unique<std::string>(std::move(hay));
        daac_iter *it = nullptr;
        bool compact = true;
        daac_status st = daac_iter_open_compact(h_.get(), mode, DAAC_ENGINE_AUTO, reinterpret_cast<const uint8_t *>(keep->data()), keep->size(),
                                                0, nullptr, &it);
        if (st == DAAC_ERR_UNSUPPORTED) {   // (patterns of several KB: the 16-byte runs)
            compact = false;
            st = daac_iter_open(h_.get(), mode, DAAC_ENGINE_AUTO, reinterpret_cast<const uint8_t *>(keep->data()), keep->size(), 0, nullptr, &it);
        }
        if (st == DAAC_ERR_MATCH_KIND) throw PanicError(panic_msg);
        if (st != DAAC_OK) throw PanicError(std::string("device scan failed: ") + daac_last_error());
        return MatchIterator(it, std::move(keep), compact);
    }
    std::unique_ptr<daac_pma, detail::PmaDeleter> h_;
};

template <class Flavor>
class BasicBuilder {  // src/bytewise/builder.rs:21-244, src/charwise/builder.rs:20-239
public:
    BasicBuilder &match_kind(MatchKind k) { kind_ = k; return *this; }
    BasicBuilder &num_free_blocks(uint32_t n) {
        if (n < 1) throw PanicError("assertion failed: n >= 1");  // builder.rs:113 / :131
        nfb_ = n;
        return *this;
    }
    Result<BasicAhoCorasick<Flavor>> build(const std::vector<std::string> &patterns) const { return run(patterns, nullptr); }
    Result<BasicAhoCorasick<Flavor>> build_with_values(const std::vector<std::pair<std::string, uint32_t>> &patvals) const {
        std::vector<std::string> pats;
        std::vector<uint32_t> vals;
        for (const auto &pv : patvals) { pats.push_back(pv.first); vals.push_back(pv.second); }
        return run(pats, &vals);
    }

private:
    Result<BasicAhoCorasick<Flavor>> run(const std::vector<std::string> &pats, const std::vector<uint32_t> *vals) const {
        std::string blob;
        std::vector<uint64_t> offs{0};
        for (const auto &p : pats) { blob += p; offs.push_back(blob.size()); }
        daac_pma *h = nullptr;
        const daac_status st = Flavor::build(reinterpret_cast<const uint8_t *>(blob.data()), offs.data(), vals ? vals->data() : nullptr,
                                             pats.size(), static_cast<uint8_t>(kind_), nfb_, &h);
        if (st != DAAC_OK) return detail::make_error(st);
        return BasicAhoCorasick<Flavor>(h);
    }
    MatchKind kind_ = MatchKind::Standard;
    uint32_t nfb_ = 16;
};

using DoubleArrayAhoCorasick = BasicAhoCorasick<detail::Bytewise>;
using DoubleArrayAhoCorasickBuilder = BasicBuilder<detail::Bytewise>;
using CharwiseDoubleArrayAhoCorasick = BasicAhoCorasick<detail::Charwise>;
using CharwiseDoubleArrayAhoCorasickBuilder = BasicBuilder<detail::Charwise>;

}  // namespace daachorse
