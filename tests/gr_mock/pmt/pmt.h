// tests/gr_mock: symbols and doubles are all the wrappers need of pmt
#pragma once
#include <boost/shared_ptr.hpp>
#include <map>
#include <stdexcept>
#include <string>
namespace pmt {
struct pmt_base {
    enum kind_t { SYMBOL, DOUBLE } kind;
    std::string sym;
    double dbl;
};
typedef boost::shared_ptr<pmt_base> pmt_t;
inline pmt_t intern(const std::string& s)
{
    static std::map<std::string, pmt_t> table; // interned: one object per spelling
    pmt_t& p = table[s];
    if (!p) {
        p.reset(new pmt_base());
        p->kind = pmt_base::SYMBOL;
        p->sym = s;
        p->dbl = 0;
    }
    return p;
}
inline pmt_t string_to_symbol(const std::string& s) { return intern(s); }
inline std::string symbol_to_string(const pmt_t& p) { return p->sym; }
inline pmt_t from_double(double x)
{
    pmt_t p(new pmt_base());
    p->kind = pmt_base::DOUBLE;
    p->dbl = x;
    return p;
}
inline double to_double(const pmt_t& p)
{
    if (!p || p->kind != pmt_base::DOUBLE)
        throw std::invalid_argument("pmt::to_double: wrong type");
    return p->dbl;
}
inline bool eq(const pmt_t& a, const pmt_t& b) { return a.get() == b.get(); }
inline bool eqv(const pmt_t& a, const pmt_t& b) { return a.get() == b.get() || (a && b && a->kind == pmt_base::DOUBLE && b->kind == pmt_base::DOUBLE && a->dbl == b->dbl); }
} // namespace pmt
