#include "json-value.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <sstream>

namespace json {

Value const * Value::get(std::string const & key) const
{
    if (type != Type::object)
        return nullptr;
    for (auto const & kv : object)
        if (kv.first == key)
            return &kv.second;
    return nullptr;
}

namespace {

class Parser
{
public:
    explicit Parser(std::string const & s) : s_(s) {}

    Value document()
    {
        Value v = value(0);
        skip_ws();
        if (pos_ != s_.size())
            fail("unexpected trailing characters");
        return v;
    }

private:
    std::string const & s_;
    size_t pos_ = 0;

    [[noreturn]] void fail(std::string const & what) const
    {
        size_t line = 1, col = 1;
        for (size_t i = 0; i < pos_ && i < s_.size(); ++i) {
            if (s_[i] == '\n') {
                ++line;
                col = 1;
            } else {
                ++col;
            }
        }
        std::ostringstream o;
        o << "line " << line << ", column " << col << ": " << what;
        throw parse_error(o.str());
    }

    void skip_ws()
    {
        while (pos_ < s_.size() &&
               (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r'))
            ++pos_;
    }

    bool consume(char c)
    {
        skip_ws();
        if (pos_ < s_.size() && s_[pos_] == c) {
            ++pos_;
            return true;
        }
        return false;
    }

    void expect_word(char const * w)
    {
        for (char const * p = w; *p; ++p, ++pos_)
            if (pos_ >= s_.size() || s_[pos_] != *p)
                fail(std::string("expected \"") + w + "\"");
    }

    static void append_utf8(std::string & out, unsigned cp)
    {
        if (cp < 0x80) {
            out += (char) cp;
        } else if (cp < 0x800) {
            out += (char) (0xC0 | (cp >> 6));
            out += (char) (0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char) (0xE0 | (cp >> 12));
            out += (char) (0x80 | ((cp >> 6) & 0x3F));
            out += (char) (0x80 | (cp & 0x3F));
        } else {
            out += (char) (0xF0 | (cp >> 18));
            out += (char) (0x80 | ((cp >> 12) & 0x3F));
            out += (char) (0x80 | ((cp >> 6) & 0x3F));
            out += (char) (0x80 | (cp & 0x3F));
        }
    }

    unsigned hex4()
    {
        unsigned v = 0;
        for (int i = 0; i < 4; ++i, ++pos_) {
            if (pos_ >= s_.size())
                fail("unterminated \\u escape");
            char c = s_[pos_];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned) (c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned) (c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned) (c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }

    std::string string_literal()
    {
        // opening quote already consumed
        std::string out;
        for (;;) {
            if (pos_ >= s_.size())
                fail("unterminated string");
            char c = s_[pos_++];
            if (c == '"')
                return out;
            if (c != '\\') {
                out += c;
                continue;
            }
            if (pos_ >= s_.size())
                fail("unterminated escape");
            char e = s_[pos_++];
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp <= 0xDBFF && pos_ + 1 < s_.size() && s_[pos_] == '\\' &&
                    s_[pos_ + 1] == 'u') {
                    pos_ += 2;
                    unsigned lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                append_utf8(out, cp);
                break;
            }
            default: --pos_; fail("unknown escape");
            }
        }
    }

    Value number()
    {
        size_t start = pos_;
        bool integral = true;
        if (pos_ < s_.size() && s_[pos_] == '-')
            ++pos_;
        if (pos_ >= s_.size() || !(s_[pos_] >= '0' && s_[pos_] <= '9'))
            fail("expected a value");
        while (pos_ < s_.size() && s_[pos_] >= '0' && s_[pos_] <= '9')
            ++pos_;
        if (pos_ < s_.size() && s_[pos_] == '.') {
            integral = false;
            ++pos_;
            while (pos_ < s_.size() && s_[pos_] >= '0' && s_[pos_] <= '9')
                ++pos_;
        }
        if (pos_ < s_.size() && (s_[pos_] == 'e' || s_[pos_] == 'E')) {
            integral = false;
            ++pos_;
            if (pos_ < s_.size() && (s_[pos_] == '+' || s_[pos_] == '-'))
                ++pos_;
            while (pos_ < s_.size() && s_[pos_] >= '0' && s_[pos_] <= '9')
                ++pos_;
        }
        std::string tok = s_.substr(start, pos_ - start);
        Value v;
        v.type = Type::number;
        v.number = std::strtod(tok.c_str(), nullptr);
        if (integral) {
            errno = 0;
            long long i = std::strtoll(tok.c_str(), nullptr, 10);
            if (errno == 0) {
                v.is_integer = true;
                v.integer = i;
            }
        }
        return v;
    }

    Value value(int depth)
    {
        if (depth > 256)
            fail("nesting too deep");
        skip_ws();
        if (pos_ >= s_.size())
            fail("unexpected end of input");
        Value v;
        char c = s_[pos_];
        if (c == '{') {
            ++pos_;
            v.type = Type::object;
            if (consume('}'))
                return v;
            for (;;) {
                if (!consume('"'))
                    fail("expected a member name");
                std::string key = string_literal();
                if (!consume(':'))
                    fail("expected ':'");
                Value member = value(depth + 1);
                if (!v.get(key))
                    v.object.emplace_back(std::move(key), std::move(member));
                if (consume(','))
                    continue;
                if (consume('}'))
                    return v;
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++pos_;
            v.type = Type::array;
            if (consume(']'))
                return v;
            for (;;) {
                v.array.push_back(value(depth + 1));
                if (consume(','))
                    continue;
                if (consume(']'))
                    return v;
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            ++pos_;
            v.type = Type::string;
            v.string = string_literal();
            return v;
        }
        if (c == 't') {
            expect_word("true");
            v.type = Type::boolean;
            v.boolean = true;
            return v;
        }
        if (c == 'f') {
            expect_word("false");
            v.type = Type::boolean;
            return v;
        }
        if (c == 'n') {
            expect_word("null");
            return v;
        }
        return number();
    }
};

} // namespace

Value parse(std::string const & text)
{
    return Parser(text).document();
}

} // namespace json
