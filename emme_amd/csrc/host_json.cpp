// host_json.cpp -- see host_json.hpp for the grammar contract and reference citations.
#include "host_json.hpp"

#include <cstdlib>
#include <sstream>

namespace emme {

namespace {

const char* kind_name(JsonValue::Kind k) {
    // spelled like the reference's category names so error texts stay comparable
    switch (k) {
        case JsonValue::Null: return "ValueCategory::Null";
        case JsonValue::Int: return "ValueCategory::NumberInt";
        case JsonValue::Float: return "ValueCategory::NumberFloat";
        case JsonValue::Bool: return "ValueCategory::Boolean";
        case JsonValue::String: return "ValueCategory::String";
        case JsonValue::Array: return "ValueCategory::Array";
        case JsonValue::Object: return "ValueCategory::Object";
    }
    return "";
}

[[noreturn]] void type_error(const char* wanted, JsonValue::Kind got) {
    if (got == JsonValue::Null) throw std::runtime_error("Undefined Property");
    std::ostringstream o;
    o << "Incorrect JSON type, requires" << wanted << "actually: " << kind_name(got);
    throw std::runtime_error(o.str());
}

struct Cursor {
    const std::string& t;
    const std::string& file;
    size_t pos = 0;
    int row = 1, col = 1;

    [[noreturn]] void fail(const std::string& what, int r, int c) const {
        std::ostringstream o;
        o << file << ':' << r << ':' << c << ": error: " << what;
        throw std::runtime_error(o.str());
    }
    bool eof() const { return pos >= t.size(); }
    char peek() const { return t[pos]; }
    char take() {
        char c = t[pos++];
        if (c == '\n') {
            ++row;
            col = 1;
        } else {
            ++col;
        }
        return c;
    }
    void skip_ws() {
        while (!eof()) {
            char c = peek();
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r')
                take();
            else
                break;
        }
    }
};

bool num_start(char c) { return (c >= '0' && c <= '9') || c == '-' || c == '+'; }
bool num_body(char c) { return num_start(c) || c == '.' || c == 'e' || c == 'E'; }

JsonValue parse_value(Cursor& c);

JsonValue parse_number(Cursor& c) {
    std::string tok;
    bool has_dot = false;
    while (!c.eof() && num_body(c.peek())) {
        char ch = c.take();
        has_dot |= ch == '.';
        tok.push_back(ch);
    }
    JsonValue v;
    if (has_dot) {
        v.kind = JsonValue::Float;
        v.f = std::atof(tok.c_str());
    } else {
        v.kind = JsonValue::Int;  // e.g. "1e-6" -> atoi -> 1, as in the reference
        v.i = std::atoi(tok.c_str());
    }
    return v;
}

JsonValue parse_string(Cursor& c) {
    JsonValue v;
    v.kind = JsonValue::String;
    c.take();  // opening quote
    while (!c.eof() && c.peek() != '"') v.s.push_back(c.take());
    if (!c.eof()) c.take();
    return v;
}

JsonValue parse_word(Cursor& c) {
    const int r = c.row, col = c.col;
    auto expect = [&](const char* w) {
        for (const char* p = w; *p; ++p) {
            if (c.eof() || c.peek() != *p) c.fail("unrecognized token", r, col);
            c.take();
        }
    };
    JsonValue v;
    if (c.peek() == 't') {
        expect("true");
        v.kind = JsonValue::Bool;
        v.b = true;
    } else if (c.peek() == 'f') {
        expect("false");
        v.kind = JsonValue::Bool;
        v.b = false;
    } else {
        expect("null");
    }
    return v;
}

JsonValue parse_object(Cursor& c) {
    JsonValue v = JsonValue::make_object();
    c.take();  // {
    c.skip_ws();
    if (!c.eof() && c.peek() == '}') {
        c.take();
        return v;
    }
    for (;;) {
        c.skip_ws();
        if (c.eof() || c.peek() != '"')
            c.fail(std::string("unexpected content '") + (c.eof() ? "" : std::string(1, c.peek())) + "'",
                   c.row, c.col);
        JsonValue key = parse_string(c);
        c.skip_ws();
        if (c.eof() || c.peek() != ':')
            c.fail(std::string("unexpected content '") + (c.eof() ? "" : std::string(1, c.peek())) + "'",
                   c.row, c.col);
        c.take();
        JsonValue val = parse_value(c);
        if (!v.has(key.s)) v.members.emplace_back(key.s, std::move(val));  // first wins
        c.skip_ws();
        if (c.eof()) c.fail("unexpected content ''", c.row, c.col);
        char d = c.take();
        if (d == '}') break;
        if (d != ',') c.fail(std::string("unexpected content '") + d + "'", c.row, c.col - 1);
    }
    return v;
}

JsonValue parse_array(Cursor& c) {
    JsonValue v = JsonValue::make_array();
    c.take();  // [
    c.skip_ws();
    if (!c.eof() && c.peek() == ']') {
        c.take();
        return v;
    }
    for (;;) {
        v.items.push_back(parse_value(c));
        c.skip_ws();
        if (c.eof()) c.fail("unexpected content ''", c.row, c.col);
        char d = c.take();
        if (d == ']') break;
        if (d != ',') c.fail(std::string("unexpected content '") + d + "'", c.row, c.col - 1);
    }
    return v;
}

JsonValue parse_value(Cursor& c) {
    c.skip_ws();
    if (c.eof()) c.fail("unexpected content ''", c.row, c.col);
    char ch = c.peek();
    if (ch == '{') return parse_object(c);
    if (ch == '[') return parse_array(c);
    if (ch == '"') return parse_string(c);
    if (num_start(ch)) return parse_number(c);
    if (ch == 't' || ch == 'f' || ch == 'n') return parse_word(c);
    if (ch == '}' || ch == ']' || ch == ':' || ch == ',')
        c.fail(std::string("unexpected content '") + ch + "'", c.row, c.col);
    c.fail("unrecognized token", c.row, c.col);
}

}  // namespace

bool JsonValue::has(const std::string& key) const {
    for (auto& m : members)
        if (m.first == key) return true;
    return false;
}

const JsonValue& JsonValue::at(const std::string& key) const {
    if (kind == Object)
        for (auto& m : members)
            if (m.first == key) return m.second;
    throw std::runtime_error("Failed to accessing key: " + key);
}

JsonValue& JsonValue::operator[](const std::string& key) {
    if (kind != Object) type_error(": ValueCategory::Object, ", kind);
    for (auto& m : members)
        if (m.first == key) return m.second;
    members.emplace_back(key, JsonValue{});
    return members.back().second;
}

const JsonValue& JsonValue::at(size_t idx) const {
    if (kind != Array || idx >= items.size()) {
        std::ostringstream o;
        o << "Failed to accessing index: " << idx;
        throw std::runtime_error(o.str());
    }
    return items[idx];
}

double JsonValue::number() const {
    if (kind == Float) return f;
    if (kind == Int) return i;
    type_error(" one of: ValueCategory::NumberFloat, ValueCategory::NumberInt, ", kind);
}

const std::string& JsonValue::str() const {
    if (kind != String) type_error(": ValueCategory::String, ", kind);
    return s;
}

bool JsonValue::boolean() const {
    if (kind != Bool) type_error(": ValueCategory::Boolean, ", kind);
    return b;
}

std::string JsonValue::dump(int indent) const {
    std::ostringstream o;
    auto pad = [&](int n) { o << std::string((size_t)n, ' '); };
    switch (kind) {
        case Null: o << "null"; break;
        case Bool: o << (b ? "true" : "false"); break;
        case Int: o << i; break;
        case Float: o << f; break;
        case String: o << '"' << s << '"'; break;
        case Object:
            if (members.empty()) {
                o << "{ }";
                break;
            }
            o << "{\n";
            for (size_t k = 0; k < members.size(); ++k) {
                pad(indent + 4);
                o << '"' << members[k].first << "\": " << members[k].second.dump(indent + 4);
                o << (k + 1 < members.size() ? ",\n" : "\n");
            }
            pad(indent);
            o << '}';
            break;
        case Array:
            if (items.empty()) {
                o << "[ ]";
                break;
            }
            o << "[\n";
            for (size_t k = 0; k < items.size(); ++k) {
                pad(indent + 4);
                o << items[k].dump(indent + 4) << (k + 1 < items.size() ? ",\n" : "\n");
            }
            pad(indent);
            o << ']';
            break;
    }
    return o.str();
}

JsonValue json_parse(const std::string& text, const std::string& filename) {
    Cursor c{text, filename};
    JsonValue v = parse_value(c);
    c.skip_ws();
    if (!c.eof())
        c.fail(std::string("unexpected content '") + c.peek() + "'", c.row, c.col);
    return v;
}

}  // namespace emme
