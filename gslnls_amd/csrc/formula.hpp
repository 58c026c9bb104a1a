// formula.hpp -- model lowering at the boundary: R formula right-hand side -> device row model.
//
// In the reference the model closure is built from the formula,
//   .fn <- function(par, .data = mf) eval(formula[[3]], c(as.list(par), .data))   (R/nls.R:565)
// and evaluated with Rf_eval on every call (src/nls.c:836-837).  The R shim deparses
// formula[[3]] and hands the text to gslnls_lower_formula(); the expression is parsed with R's
// operator precedence (^ and ** right-associative and above unary minus, then * /, then + -)
// and matched structurally -- up to a consistent renaming of parameters and data columns --
// against the formulas of the registered device models (models.hpp).  Host-only code.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

namespace gslnls
{

struct FNode
{
    enum Kind
    {
        NUM,
        SYM,
        NEG,
        BIN,
        CALL
    } kind;
    double num = 0.0;
    std::string name; // symbol, operator or function name
    std::vector<std::shared_ptr<FNode>> args;
};
using FNodeP = std::shared_ptr<FNode>;

struct FParser
{
    struct Tok
    {
        char kind; // 'n' number, 's' symbol, 'o' operator/paren, 'e' end
        double num;
        std::string text;
    };
    std::vector<Tok> toks;
    size_t pos = 0;
    bool ok = true;

    explicit FParser(const char *src)
    {
        const char *s = src;
        while (*s)
        {
            if (*s == ' ' || *s == '\t' || *s == '\n')
            {
                ++s;
                continue;
            }
            if ((*s >= '0' && *s <= '9') || (*s == '.' && s[1] >= '0' && s[1] <= '9'))
            {
                char *end = nullptr;
                const double v = strtod(s, &end);
                toks.push_back({'n', v, ""});
                s = end;
                continue;
            }
            if ((*s >= 'A' && *s <= 'Z') || (*s >= 'a' && *s <= 'z') || *s == '_' || *s == '.')
            {
                const char *b = s;
                while ((*s >= 'A' && *s <= 'Z') || (*s >= 'a' && *s <= 'z') || (*s >= '0' && *s <= '9') || *s == '_' ||
                       *s == '.')
                    ++s;
                toks.push_back({'s', 0.0, std::string(b, s)});
                continue;
            }
            if (*s == '*' && s[1] == '*')
            {
                toks.push_back({'o', 0.0, "^"});
                s += 2;
                continue;
            }
            if (strchr("+-*/^(),", *s))
            {
                toks.push_back({'o', 0.0, std::string(1, *s)});
                ++s;
                continue;
            }
            ok = false;
            return;
        }
        toks.push_back({'e', 0.0, ""});
    }
    bool is_op(const char *t) const { return toks[pos].kind == 'o' && toks[pos].text == t; }

    FNodeP expr()
    {
        FNodeP n = term();
        while (ok && (is_op("+") || is_op("-")))
        {
            auto b = std::make_shared<FNode>();
            b->kind = FNode::BIN;
            b->name = toks[pos++].text;
            b->args = {n, term()};
            n = b;
        }
        return n;
    }
    FNodeP term()
    {
        FNodeP n = unary();
        while (ok && (is_op("*") || is_op("/")))
        {
            auto b = std::make_shared<FNode>();
            b->kind = FNode::BIN;
            b->name = toks[pos++].text;
            b->args = {n, unary()};
            n = b;
        }
        return n;
    }
    FNodeP unary()
    {
        if (is_op("-"))
        {
            ++pos;
            auto b = std::make_shared<FNode>();
            b->kind = FNode::NEG;
            b->args = {unary()};
            return b;
        }
        if (is_op("+"))
        {
            ++pos;
            return unary();
        }
        return power();
    }
    FNodeP unary_pow()
    {
        if (is_op("-"))
        {
            ++pos;
            auto b = std::make_shared<FNode>();
            b->kind = FNode::NEG;
            b->args = {unary_pow()};
            return b;
        }
        return power();
    }
    FNodeP power()
    {
        FNodeP base = atom();
        if (ok && is_op("^"))
        {
            ++pos;
            auto b = std::make_shared<FNode>();
            b->kind = FNode::BIN;
            b->name = "^";
            b->args = {base, unary_pow()};
            return b;
        }
        return base;
    }
    FNodeP atom()
    {
        auto n = std::make_shared<FNode>();
        const Tok t = toks[pos];
        if (t.kind == 'n')
        {
            ++pos;
            n->kind = FNode::NUM;
            n->num = t.num;
            return n;
        }
        if (t.kind == 's')
        {
            ++pos;
            if (is_op("("))
            {
                ++pos;
                n->kind = FNode::CALL;
                n->name = t.text;
                if (!is_op(")"))
                {
                    n->args.push_back(expr());
                    while (ok && is_op(","))
                    {
                        ++pos;
                        n->args.push_back(expr());
                    }
                }
                if (!is_op(")"))
                    ok = false;
                else
                    ++pos;
                return n;
            }
            n->kind = FNode::SYM;
            n->name = t.text;
            return n;
        }
        if (is_op("("))
        {
            ++pos;
            FNodeP e = expr();
            if (!is_op(")"))
                ok = false;
            else
                ++pos;
            return e;
        }
        ok = false;
        n->kind = FNode::NUM;
        return n;
    }
    FNodeP parse()
    {
        if (!ok)
            return nullptr;
        FNodeP e = expr();
        if (!ok || toks[pos].kind != 'e')
            return nullptr;
        return e;
    }
};

inline bool fmatch(const FNodeP &t, const FNodeP &u, std::map<std::string, std::string> &pmap,
                   std::map<std::string, std::string> &dmap, const std::set<std::string> &tpar,
                   const std::set<std::string> &upar)
{
    if (t->kind != u->kind || t->args.size() != u->args.size())
        return false;
    switch (t->kind)
    {
    case FNode::NUM:
        return t->num == u->num;
    case FNode::SYM:
    {
        const bool tp = tpar.count(t->name) > 0, up = upar.count(u->name) > 0;
        if (tp != up)
            return false;
        auto &m = tp ? pmap : dmap;
        auto it = m.find(t->name);
        if (it != m.end())
            return it->second == u->name;
        for (auto &kv : m)
            if (kv.second == u->name)
                return false;
        m[t->name] = u->name;
        return true;
    }
    case FNode::BIN:
    case FNode::CALL:
        if (t->name != u->name)
            return false;
        break;
    default:
        break;
    }
    for (size_t i = 0; i < t->args.size(); ++i)
        if (!fmatch(t->args[i], u->args[i], pmap, dmap, tpar, upar))
            return false;
    return true;
}

struct FRegistryEntry
{
    int id;
    const char *rhs;
    std::vector<std::string> par, var;
};

inline const std::vector<FRegistryEntry> &fregistry()
{
    static const std::vector<FRegistryEntry> reg = {
        {1, "A*exp(-lam*x)+b", {"A", "lam", "b"}, {"x"}},
        {2, "b1*(1-exp(-b2*x))", {"b1", "b2"}, {"x"}},
        {3, "a*exp(-(x-b)^2/(2*c^2))", {"a", "b", "c"}, {"x"}},
        {4,
         "b1*exp(-b2*x) + b3*exp(-(x-b4)^2/b5^2) + b6*exp(-(x-b7)^2/b8^2)",
         {"b1", "b2", "b3", "b4", "b5", "b6", "b7", "b8"},
         {"x"}},
    };
    return reg;
}

// returns the registry id (> 0) or 0 when the expression does not lower; par_order[k] = index into
// parnames of the k-th device parameter; var_names_out[c] = data column feeding device regressor c
inline int lower_formula(const char *rhs, int p, const char *const *parnames, int *par_order,
                         std::vector<std::string> &var_names_out)
{
    FParser up(rhs);
    FNodeP u = up.parse();
    if (!u)
        return 0;
    std::set<std::string> upar;
    for (int k = 0; k < p; ++k)
        upar.insert(parnames[k]);
    for (const auto &e : fregistry())
    {
        if ((int)e.par.size() != p)
            continue;
        FParser tp(e.rhs);
        FNodeP t = tp.parse();
        std::map<std::string, std::string> pmap, dmap;
        std::set<std::string> tpar(e.par.begin(), e.par.end());
        if (!fmatch(t, u, pmap, dmap, tpar, upar) || (int)pmap.size() != p)
            continue;
        for (int k = 0; k < p; ++k)
        {
            const std::string &un = pmap[e.par[k]];
            for (int j = 0; j < p; ++j)
                if (un == parnames[j])
                    par_order[k] = j;
        }
        var_names_out.clear();
        for (const auto &v : e.var)
            var_names_out.push_back(dmap[v]);
        return e.id;
    }
    return 0;
}

} // namespace gslnls
