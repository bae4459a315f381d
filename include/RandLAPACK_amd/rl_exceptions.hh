// RandLAPACK::Error + randlapack_require, same contract as the reference's RandLAPACK/rl_exceptions.hh:37-52,97-98
// (stream a message after the macro; a failed requirement throws RandLAPACK::Error).
#pragma once
#include <exception>
#include <sstream>
#include <string>

namespace RandLAPACK {

class Error : public std::exception {
    std::string text_;
public:
    Error() = default;
    explicit Error(std::string const& msg) : text_(msg) {}
    Error(std::string const& msg, const char* func) : text_(msg + ", in function " + func) {}
    const char* what() const noexcept override { return text_.c_str(); }
};

namespace exceptions::internal {
// Built only on the failing path; throws when the full expression ends.
struct Thrower {
    std::ostringstream os;
    const char* cond;
    const char* func;
    bool touched = false;
    Thrower(const char* c, const char* f) : cond(c), func(f) {}
    template <typename X>
    Thrower& operator<<(X const& x) { os << x; touched = true; return *this; }
    ~Thrower() noexcept(false) {
        std::string msg = touched ? os.str() : (std::string("requirement failed: ") + cond);
        throw Error(msg, func);
    }
};
}  // namespace exceptions::internal
}  // namespace RandLAPACK

#define randlapack_require(cond) \
    if (cond) {} else ::RandLAPACK::exceptions::internal::Thrower(#cond, __func__)
