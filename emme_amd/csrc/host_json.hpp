// host_json.hpp -- small JSON reader/writer for the host layer.
//
// Accepts the grammar of the reference's hand-written parser and keeps its observable
// quirks (reference src/JsonParser.cpp:381-491 lexer, :521-653 parser):
//   * a number token starts with [0-9+-] and continues over [0-9+-.eE]; it is a FLOAT only
//     if it contains '.', otherwise an INTEGER converted with atoi -- so `1e-6` reads as 1
//     while `1.e-6` / `1.0e-6` read as 1e-6 (:436-446, :558-570);
//   * strings have no escape sequences (:429-432);
//   * on duplicate keys the first one wins (unordered_map::emplace, :593);
//   * missing key -> "Failed to accessing key: <k>" (:98-106).
// The data model is this project's own (tagged struct + ordered member list).
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace emme {

struct JsonValue {
    enum Kind { Null, Int, Float, Bool, String, Array, Object } kind = Null;
    int i = 0;
    double f = 0.0;
    bool b = false;
    std::string s;
    std::vector<JsonValue> items;                            // Array
    std::vector<std::pair<std::string, JsonValue>> members;  // Object (insertion order)

    bool is_object() const { return kind == Object; }
    bool is_array() const { return kind == Array; }
    bool is_string() const { return kind == String; }
    bool has(const std::string& key) const;
    const JsonValue& at(const std::string& key) const;  // throws like the reference
    JsonValue& operator[](const std::string& key);      // creates a Null member if absent
    const JsonValue& at(size_t idx) const;
    double number() const;  // Int or Float, else "Incorrect JSON type" error
    const std::string& str() const;
    bool boolean() const;

    static JsonValue make_object() { JsonValue v; v.kind = Object; return v; }
    static JsonValue make_array() { JsonValue v; v.kind = Array; return v; }
    static JsonValue make(double x) { JsonValue v; v.kind = Float; v.f = x; return v; }
    static JsonValue make(int x) { JsonValue v; v.kind = Int; v.i = x; return v; }
    static JsonValue make(const std::string& x) { JsonValue v; v.kind = String; v.s = x; return v; }
    static JsonValue make(bool x) { JsonValue v; v.kind = Bool; v.b = x; return v; }

    // Text output; doubles use the default ostream precision (6 significant digits) like
    // the reference (src/JsonParser.cpp:227).
    std::string dump(int indent = 0) const;
};

JsonValue json_parse(const std::string& text, const std::string& filename = "<string>");

}  // namespace emme
