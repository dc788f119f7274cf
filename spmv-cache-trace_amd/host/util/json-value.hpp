// json-value.hpp -- a small JSON document model and parser.
//
// Only the trace configuration is read with it (the role src/util/json.{h,c} plays in the
// reference).  Objects keep their members in file order; duplicate keys keep the first.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace json {

class parse_error : public std::runtime_error
{
public:
    explicit parse_error(std::string const & what) : std::runtime_error(what) {}
};

enum class Type { null, boolean, number, string, array, object };

class Value
{
public:
    Type type = Type::null;
    bool boolean = false;
    double number = 0.0;
    bool is_integer = false; // the literal had no fraction or exponent
    long long integer = 0;
    std::string string;
    std::vector<Value> array;
    std::vector<std::pair<std::string, Value>> object;

    bool is_null() const { return type == Type::null; }
    bool is_number() const { return type == Type::number; }
    bool is_string() const { return type == Type::string; }
    bool is_array() const { return type == Type::array; }
    bool is_object() const { return type == Type::object; }

    // member lookup; nullptr when absent or when this is not an object
    Value const * get(std::string const & key) const;
    long long to_int() const { return is_integer ? integer : (long long) number; }
};

// Parses one JSON document (trailing whitespace allowed); throws parse_error with a
// "line L, column C: ..." message.
Value parse(std::string const & text);

} // namespace json
