// json-ostreambuf.hpp -- output filter that pretty-prints the JSON the program emits.
//
// Same observable behaviour as the reference's json_ostreambuf
// (src/util/json-ostreambuf.cpp:76-89 and the num_put facet :16-28): the producers write
// compact JSON with '\n' where they want line breaks; this buffer indents each line by two
// spaces per open '{' / '[' and prints a NaN double as the string "nan".  Installed on a
// stream by construction, removed by destruction.
#pragma once

#include <cmath>
#include <locale>
#include <ostream>
#include <streambuf>
#include <string>

class json_ostreambuf : public std::streambuf
{
public:
    explicit json_ostreambuf(std::ostream & stream)
        : stream_(stream), sink_(stream.rdbuf()), saved_locale_(stream.getloc())
    {
        stream_.rdbuf(this);
        stream_.imbue(std::locale(saved_locale_, new nan_as_string));
    }

    ~json_ostreambuf() override
    {
        stream_.rdbuf(sink_);
        stream_.imbue(saved_locale_);
    }

    json_ostreambuf(json_ostreambuf const &) = delete;
    json_ostreambuf & operator=(json_ostreambuf const &) = delete;

protected:
    int_type overflow(int_type ch) override
    {
        if (traits_type::eq_int_type(ch, traits_type::eof()))
            return traits_type::not_eof(ch);
        char const c = traits_type::to_char_type(ch);
        if (c == '}' || c == ']')
            depth_ -= 2;
        if (at_line_start_ && c != '\n')
            for (int i = 0; i < depth_; ++i)
                sink_->sputc(' ');
        at_line_start_ = (c == '\n');
        if (c == '{' || c == '[')
            depth_ += 2;
        return sink_->sputc(c);
    }

    int sync() override { return sink_->pubsync(); }

private:
    // doubles print as usual, except NaN -> "nan" (a JSON string, so the document stays valid)
    class nan_as_string : public std::num_put<char>
    {
    protected:
        iter_type do_put(iter_type out, std::ios_base & s, char_type fill, double v) const override
        {
            return std::isnan(v) ? put_nan(out) : std::num_put<char>::do_put(out, s, fill, v);
        }
        iter_type do_put(iter_type out, std::ios_base & s, char_type fill, long double v) const override
        {
            return std::isnan(v) ? put_nan(out) : std::num_put<char>::do_put(out, s, fill, v);
        }

    private:
        static iter_type put_nan(iter_type out)
        {
            for (char c : std::string("\"nan\""))
                *out++ = c;
            return out;
        }
    };

    std::ostream & stream_;
    std::streambuf * sink_;
    std::locale saved_locale_;
    bool at_line_start_ = true;
    int depth_ = 0;
};
