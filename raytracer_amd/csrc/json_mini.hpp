// json_mini.hpp -- a small recursive-descent JSON reader.
//
// The reference parses its scene files with nlohmann::json
// (reference src/scene.cu:1,19,93); only the subset its scenes use is needed:
// objects, arrays, numbers, strings, booleans, null.  Numbers are read with
// strtod and narrowed by the caller, which is what nlohmann's
// double -> float conversion does.
#pragma once

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rt {
namespace json {

class Value {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };

    Kind kind = Null;
    bool boolean = false;
    double number = 0.0;
    std::string text;
    std::vector<Value> items;
    std::vector<std::pair<std::string, Value>> members;

    // Element count of an array or object; 0 for anything else, so a missing
    // key iterates zero times like `jf["objects"].size()` on a null value.
    size_t size() const {
        if (kind == Array) return items.size();
        if (kind == Object) return members.size();
        return 0;
    }

    // Member lookup; a shared Null value when absent or not an object.
    const Value& operator[](const std::string& key) const {
        if (kind == Object)
            for (const auto& m : members)
                if (m.first == key) return m.second;
        return null_value();
    }

    const Value& operator[](size_t i) const {
        if (kind != Array || i >= items.size()) throw std::runtime_error("json: array index out of range");
        return items[i];
    }

    bool is_null() const { return kind == Null; }

    double as_double() const {
        if (kind != Number) throw std::runtime_error("json: number expected");
        return number;
    }
    float as_float() const { return static_cast<float>(as_double()); }
    const std::string& as_string() const {
        if (kind != String) throw std::runtime_error("json: string expected");
        return text;
    }

private:
    static const Value& null_value() {
        static const Value v;
        return v;
    }
};

class Parser {
public:
    explicit Parser(const std::string& s) : src(s) {}

    Value parse_document() {
        Value v = parse_value();
        skip_space();
        if (pos != src.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& src;
    size_t pos = 0;

    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(pos));
    }

    void skip_space() {
        while (pos < src.size() && (src[pos] == ' ' || src[pos] == '\t' || src[pos] == '\n' || src[pos] == '\r')) pos++;
    }

    bool consume(char c) {
        skip_space();
        if (pos < src.size() && src[pos] == c) { pos++; return true; }
        return false;
    }

    void expect_word(const char* w) {
        for (const char* p = w; *p; ++p, ++pos)
            if (pos >= src.size() || src[pos] != *p) fail("bad literal");
    }

    std::string parse_string() {
        std::string out;
        pos++; // opening quote
        while (true) {
            if (pos >= src.size()) fail("unterminated string");
            char c = src[pos++];
            if (c == '"') break;
            if (c != '\\') { out.push_back(c); continue; }
            if (pos >= src.size()) fail("unterminated escape");
            char e = src[pos++];
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (pos + 4 > src.size()) fail("short \\u escape");
                    unsigned code = static_cast<unsigned>(std::strtoul(src.substr(pos, 4).c_str(), nullptr, 16));
                    pos += 4;
                    // UTF-8 encode the BMP code point (surrogate pairs are not joined)
                    if (code < 0x80) out.push_back(static_cast<char>(code));
                    else if (code < 0x800) {
                        out.push_back(static_cast<char>(0xC0 | (code >> 6)));
                        out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
                    } else {
                        out.push_back(static_cast<char>(0xE0 | (code >> 12)));
                        out.push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
                        out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
                    }
                    break;
                }
                default: fail("unknown escape");
            }
        }
        return out;
    }

    Value parse_value() {
        skip_space();
        if (pos >= src.size()) fail("unexpected end");
        Value v;
        char c = src[pos];
        if (c == '{') {
            pos++;
            v.kind = Value::Object;
            if (consume('}')) return v;
            do {
                skip_space();
                if (pos >= src.size() || src[pos] != '"') fail("member name expected");
                std::string key = parse_string();
                if (!consume(':')) fail("':' expected");
                v.members.emplace_back(std::move(key), parse_value());
            } while (consume(','));
            if (!consume('}')) fail("'}' expected");
        } else if (c == '[') {
            pos++;
            v.kind = Value::Array;
            if (consume(']')) return v;
            do { v.items.push_back(parse_value()); } while (consume(','));
            if (!consume(']')) fail("']' expected");
        } else if (c == '"') {
            v.kind = Value::String;
            v.text = parse_string();
        } else if (c == 't') {
            expect_word("true"); v.kind = Value::Bool; v.boolean = true;
        } else if (c == 'f') {
            expect_word("false"); v.kind = Value::Bool; v.boolean = false;
        } else if (c == 'n') {
            expect_word("null");
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* begin = src.c_str() + pos;
            char* end = nullptr;
            v.number = std::strtod(begin, &end);
            if (end == begin) fail("bad number");
            pos += static_cast<size_t>(end - begin);
            v.kind = Value::Number;
        } else {
            fail("unexpected character");
        }
        return v;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse_document(); }

} // namespace json
} // namespace rt
