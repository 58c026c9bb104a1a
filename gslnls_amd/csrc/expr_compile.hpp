// expr_compile.hpp -- host compiler: formula right-hand side -> VmProgram (value + symbolic gradient).
//
// Stands in for what R does at the boundary: the model closure evaluates formula[[3]] (R/nls.R:565) and
// jac = TRUE asks stats::deriv() for the gradient expression (R/nls.R:588-599).  The AST comes from
// FParser (formula.hpp).  Nodes are hash-consed (structural CSE), constants are folded, and the usual
// algebraic identities (x+0, x*1, x*0, x^1, ...) keep the derivative DAG small.
#pragma once
#include <math.h>
#include <map>
#include <string>
#include <tuple>
#include <vector>
#include "formula.hpp"
#include "vm_program.hpp"

namespace gslnls
{

struct ExprCompiler
{
    enum Kind
    {
        K_CONST,
        K_PARAM,
        K_VAR,
        K_VDIR, // component k of the direction of the second directional derivative
        K_OP
    };
    struct Node
    {
        Kind kind;
        unsigned char op;
        int a, b;   // children (K_OP) or index (K_PARAM / K_VAR)
        double c;   // K_CONST
    };
    std::vector<Node> nodes;
    std::map<std::tuple<int, int, int, int, long long>, int> memo;
    std::string error;

    int intern(Kind k, unsigned char op, int a, int b, double c)
    {
        long long bits;
        memcpy(&bits, &c, sizeof(bits));
        auto key = std::make_tuple((int)k, (int)op, a, b, bits);
        auto it = memo.find(key);
        if (it != memo.end())
            return it->second;
        nodes.push_back({k, op, a, b, c});
        memo[key] = (int)nodes.size() - 1;
        return (int)nodes.size() - 1;
    }
    int cst(double v) { return intern(K_CONST, 0, -1, -1, v); }
    int param(int k) { return intern(K_PARAM, 0, k, -1, 0.0); }
    int var(int c) { return intern(K_VAR, 0, c, -1, 0.0); }
    int vdir(int k) { return intern(K_VDIR, 0, k, -1, 0.0); }
    bool is_const(int n) const { return nodes[n].kind == K_CONST; }
    bool is_val(int n, double v) const { return nodes[n].kind == K_CONST && nodes[n].c == v; }

    int op2(unsigned char op, int a, int b)
    {
        if (is_const(a) && is_const(b))
            return cst(vm_apply(op, nodes[a].c, nodes[b].c));
        switch (op)
        {
        case VM_ADD:
            if (is_val(a, 0.0))
                return b;
            if (is_val(b, 0.0))
                return a;
            break;
        case VM_SUB:
            if (is_val(b, 0.0))
                return a;
            if (is_val(a, 0.0))
                return op1(VM_NEG, b);
            if (a == b)
                return cst(0.0);
            break;
        case VM_MUL:
            if (is_val(a, 0.0) || is_val(b, 0.0))
                return cst(0.0);
            if (is_val(a, 1.0))
                return b;
            if (is_val(b, 1.0))
                return a;
            if (is_val(a, -1.0))
                return op1(VM_NEG, b);
            if (is_val(b, -1.0))
                return op1(VM_NEG, a);
            if (b < a)
                std::swap(a, b); // commutative: canonical order helps CSE
            break;
        case VM_DIV:
            if (is_val(a, 0.0))
                return cst(0.0);
            if (is_val(b, 1.0))
                return a;
            break;
        case VM_POW:
            if (is_val(b, 1.0))
                return a;
            if (is_val(b, 0.0))
                return cst(1.0);
            // small fixed exponents without the general pow() (R's own R_pow special-cases y == 2 the same way)
            if (is_val(b, 2.0))
                return op2(VM_MUL, a, a);
            if (is_val(b, 3.0))
                return op2(VM_MUL, op2(VM_MUL, a, a), a);
            if (is_val(b, 4.0))
            {
                const int sq = op2(VM_MUL, a, a);
                return op2(VM_MUL, sq, sq);
            }
            if (is_val(b, -1.0))
                return op2(VM_DIV, cst(1.0), a);
            if (is_val(b, -2.0))
                return op2(VM_DIV, cst(1.0), op2(VM_MUL, a, a));
            if (is_val(b, 0.5))
                return op1(VM_SQRT, a);
            break;
        default:
            break;
        }
        if (op == VM_ADD && b < a)
            std::swap(a, b);
        return intern(K_OP, op, a, b, 0.0);
    }
    int op1(unsigned char op, int a)
    {
        if (is_const(a))
            return cst(vm_apply(op, nodes[a].c, 0.0));
        if (op == VM_NEG && nodes[a].kind == K_OP && nodes[a].op == VM_NEG)
            return nodes[a].a;
        return intern(K_OP, op, a, a, 0.0);
    }

    // AST -> DAG
    int build(const FNodeP &t, const std::vector<std::string> &parnames, const std::vector<std::string> &varnames)
    {
        switch (t->kind)
        {
        case FNode::NUM:
            return cst(t->num);
        case FNode::SYM:
        {
            for (size_t k = 0; k < parnames.size(); ++k)
                if (parnames[k] == t->name)
                    return param((int)k);
            for (size_t c = 0; c < varnames.size(); ++c)
                if (varnames[c] == t->name)
                    return var((int)c);
            if (t->name == "pi")
                return cst(3.14159265358979323846);
            error = "unknown symbol '" + t->name + "'";
            return cst(NAN);
        }
        case FNode::NEG:
            return op1(VM_NEG, build(t->args[0], parnames, varnames));
        case FNode::BIN:
        {
            const int a = build(t->args[0], parnames, varnames), b = build(t->args[1], parnames, varnames);
            if (t->name == "+")
                return op2(VM_ADD, a, b);
            if (t->name == "-")
                return op2(VM_SUB, a, b);
            if (t->name == "*")
                return op2(VM_MUL, a, b);
            if (t->name == "/")
                return op2(VM_DIV, a, b);
            if (t->name == "^")
                return op2(VM_POW, a, b);
            error = "unknown operator " + t->name;
            return cst(NAN);
        }
        case FNode::CALL:
        {
            // the functions stats::deriv knows (R/nls.R:588-599 builds the Jacobian with it): primitives of the program, or
            // -- where that costs no accuracy -- compositions of them
            static const std::map<std::string, unsigned char> f1 = {
                {"exp", VM_EXP}, {"log", VM_LOG}, {"sin", VM_SIN}, {"cos", VM_COS}, {"tan", VM_TAN},
                {"atan", VM_ATAN}, {"sqrt", VM_SQRT}, {"abs", VM_ABS}, {"tanh", VM_TANH}, {"sinh", VM_SINH},
                {"cosh", VM_COSH}, {"asin", VM_ASIN}, {"acos", VM_ACOS}, {"log1p", VM_LOG1P}, {"expm1", VM_EXPM1},
                {"pnorm", VM_PNORM}, {"gamma", VM_GAMMA}, {"lgamma", VM_LGAMMA}, {"digamma", VM_PSI0}, {"trigamma", VM_PSI1}};
            // the standard selfStart models by their closed forms (stats::SSasymp & co.: the reference's unit tests 6.x fit
            // y ~ SSasymp(x, Asym, R0, lrc), inst/unit_tests/unit_tests_gslnls.R:267-293; R evaluates the model's own
            // compiled gradient attribute, here the closed form is differentiated like any other expression)
            {
                const size_t na = t->args.size();
                auto arg_n = [&](size_t k) { return build(t->args[k], parnames, varnames); };
                auto add = [&](int a, int b) { return op2(VM_ADD, a, b); };
                auto sub = [&](int a, int b) { return op2(VM_SUB, a, b); };
                auto mul = [&](int a, int b) { return op2(VM_MUL, a, b); };
                auto dvd = [&](int a, int b) { return op2(VM_DIV, a, b); };
                auto ex = [&](int a) { return op1(VM_EXP, a); };
                auto neg = [&](int a) { return op1(VM_NEG, a); };
                const std::string &f = t->name;
                if (f == "SSasymp" && na == 4) // Asym + (R0 - Asym) exp(-exp(lrc) input)
                {
                    const int in = arg_n(0), As = arg_n(1), R0 = arg_n(2), lrc = arg_n(3);
                    return add(As, mul(sub(R0, As), ex(neg(mul(ex(lrc), in)))));
                }
                if (f == "SSasympOff" && na == 4) // Asym (1 - exp(-exp(lrc) (input - c0)))
                {
                    const int in = arg_n(0), As = arg_n(1), lrc = arg_n(2), c0 = arg_n(3);
                    return mul(As, sub(cst(1.0), ex(neg(mul(ex(lrc), sub(in, c0))))));
                }
                if (f == "SSasympOrig" && na == 3) // Asym (1 - exp(-exp(lrc) input))
                {
                    const int in = arg_n(0), As = arg_n(1), lrc = arg_n(2);
                    return mul(As, sub(cst(1.0), ex(neg(mul(ex(lrc), in)))));
                }
                if (f == "SSbiexp" && na == 5) // A1 exp(-exp(lrc1) input) + A2 exp(-exp(lrc2) input)
                {
                    const int in = arg_n(0), A1 = arg_n(1), l1 = arg_n(2), A2 = arg_n(3), l2 = arg_n(4);
                    return add(mul(A1, ex(neg(mul(ex(l1), in)))), mul(A2, ex(neg(mul(ex(l2), in)))));
                }
                if (f == "SSfol" && na == 5) // Dose exp(lKe + lKa - lCl) (exp(-exp(lKe) t) - exp(-exp(lKa) t)) / (exp(lKa) - exp(lKe))
                {
                    const int D = arg_n(0), in = arg_n(1), lKe = arg_n(2), lKa = arg_n(3), lCl = arg_n(4);
                    return dvd(mul(mul(D, ex(sub(add(lKe, lKa), lCl))), sub(ex(neg(mul(ex(lKe), in))), ex(neg(mul(ex(lKa), in))))),
                               sub(ex(lKa), ex(lKe)));
                }
                if (f == "SSfpl" && na == 5) // A + (B - A) / (1 + exp((xmid - input) / scal))
                {
                    const int in = arg_n(0), A = arg_n(1), B = arg_n(2), xm = arg_n(3), sc = arg_n(4);
                    return add(A, dvd(sub(B, A), add(cst(1.0), ex(dvd(sub(xm, in), sc)))));
                }
                if (f == "SSgompertz" && na == 4) // Asym exp(-b2 b3^x)
                {
                    const int in = arg_n(0), As = arg_n(1), b2 = arg_n(2), b3 = arg_n(3);
                    return mul(As, ex(neg(mul(b2, op2(VM_POW, b3, in)))));
                }
                if (f == "SSlogis" && na == 4) // Asym / (1 + exp((xmid - input) / scal))
                {
                    const int in = arg_n(0), As = arg_n(1), xm = arg_n(2), sc = arg_n(3);
                    return dvd(As, add(cst(1.0), ex(dvd(sub(xm, in), sc))));
                }
                if (f == "SSmicmen" && na == 3) // Vm input / (K + input)
                {
                    const int in = arg_n(0), Vm = arg_n(1), K = arg_n(2);
                    return dvd(mul(Vm, in), add(K, in));
                }
                if (f == "SSweibull" && na == 5) // Asym - Drop exp(-exp(lrc) x^pwr)
                {
                    const int in = arg_n(0), As = arg_n(1), Dr = arg_n(2), lrc = arg_n(3), pw = arg_n(4);
                    return sub(As, mul(Dr, ex(neg(mul(ex(lrc), op2(VM_POW, in, pw))))));
                }
            }
            if (t->name == "psigamma" && (t->args.size() == 1 || t->args.size() == 2))
            {
                // psigamma(x, deriv = 0L): the order has to be a literal 0..4 (stats::deriv's own rule: a constant order)
                int order = 0;
                if (t->args.size() == 2)
                {
                    const int on = build(t->args[1], parnames, varnames);
                    const double ov = nodes[on].kind == K_CONST ? nodes[on].c : NAN;
                    if (!(ov == 0.0 || ov == 1.0 || ov == 2.0 || ov == 3.0 || ov == 4.0))
                    {
                        error = "psigamma: the order has to be a constant 0..4";
                        return cst(NAN);
                    }
                    order = (int)ov;
                }
                return op1((unsigned char)(VM_PSI0 + order), build(t->args[0], parnames, varnames));
            }
            if (t->args.size() != 1)
            {
                error = "unsupported function " + t->name + " (one argument expected)";
                return cst(NAN);
            }
            const int arg = build(t->args[0], parnames, varnames);
            if (t->name == "log2")
                return op2(VM_DIV, op1(VM_LOG, arg), cst(0.69314718055994530942));
            if (t->name == "log10")
                return op2(VM_DIV, op1(VM_LOG, arg), cst(2.30258509299404568402));
            if (t->name == "dnorm") // exp(-x^2 / 2) / sqrt(2 pi)
                return op2(VM_MUL, op1(VM_EXP, op2(VM_MUL, cst(-0.5), op2(VM_MUL, arg, arg))), cst(0.39894228040143267794));
            if (t->name == "factorial") // gamma(x + 1)
                return op1(VM_GAMMA, op2(VM_ADD, arg, cst(1.0)));
            if (t->name == "lfactorial") // lgamma(x + 1)
                return op1(VM_LGAMMA, op2(VM_ADD, arg, cst(1.0)));
            if (t->name == "sinpi" || t->name == "cospi" || t->name == "tanpi")
                return op1(t->name == "sinpi" ? VM_SIN : (t->name == "cospi" ? VM_COS : VM_TAN),
                           op2(VM_MUL, cst(3.14159265358979323846), arg));
            auto it = f1.find(t->name);
            if (it == f1.end())
            {
                error = "unsupported function " + t->name;
                return cst(NAN);
            }
            return op1(it->second, arg);
        }
        }
        return cst(NAN);
    }

    bool depends(int n, int k, std::map<int, bool> &cache)
    {
        auto it = cache.find(n);
        if (it != cache.end())
            return it->second;
        bool r = false;
        const Node &nd = nodes[n];
        if (nd.kind == K_PARAM)
            r = nd.a == k;
        else if (nd.kind == K_OP)
            r = depends(nd.a, k, cache) || (nd.b != nd.a && depends(nd.b, k, cache));
        cache[n] = r;
        return r;
    }

    // d node / d theta_k
    int diff(int n, int k, std::map<int, int> &dcache, std::map<int, bool> &dep)
    {
        auto it = dcache.find(n);
        if (it != dcache.end())
            return it->second;
        int r;
        const Node nd = nodes[n];
        if (!depends(n, k, dep))
            r = cst(0.0);
        else if (nd.kind == K_PARAM)
            r = cst(1.0);
        else
        {
            const int a = nd.a, b = nd.b;
            switch (nd.op)
            {
            case VM_ADD:
                r = op2(VM_ADD, diff(a, k, dcache, dep), diff(b, k, dcache, dep));
                break;
            case VM_SUB:
                r = op2(VM_SUB, diff(a, k, dcache, dep), diff(b, k, dcache, dep));
                break;
            case VM_MUL:
                r = op2(VM_ADD, op2(VM_MUL, diff(a, k, dcache, dep), b), op2(VM_MUL, a, diff(b, k, dcache, dep)));
                break;
            case VM_DIV:
            {
                // (a/b)' = a'/b - (a/b) b'/b
                const int da = diff(a, k, dcache, dep), db = diff(b, k, dcache, dep);
                r = op2(VM_SUB, op2(VM_DIV, da, b), op2(VM_DIV, op2(VM_MUL, n, db), b));
                break;
            }
            case VM_NEG:
                r = op1(VM_NEG, diff(a, k, dcache, dep));
                break;
            case VM_POW:
            {
                const bool da_dep = depends(a, k, dep), db_dep = depends(b, k, dep);
                int term = cst(0.0);
                if (da_dep) // b a^(b-1) a'
                    term = op2(VM_MUL, op2(VM_MUL, b, op2(VM_POW, a, op2(VM_SUB, b, cst(1.0)))), diff(a, k, dcache, dep));
                if (db_dep) // a^b log(a) b'
                    term = op2(VM_ADD, term, op2(VM_MUL, op2(VM_MUL, n, op1(VM_LOG, a)), diff(b, k, dcache, dep)));
                r = term;
                break;
            }
            case VM_EXP:
                r = op2(VM_MUL, n, diff(a, k, dcache, dep));
                break;
            case VM_LOG:
                r = op2(VM_DIV, diff(a, k, dcache, dep), a);
                break;
            case VM_SIN:
                r = op2(VM_MUL, op1(VM_COS, a), diff(a, k, dcache, dep));
                break;
            case VM_COS:
                r = op1(VM_NEG, op2(VM_MUL, op1(VM_SIN, a), diff(a, k, dcache, dep)));
                break;
            case VM_TAN:
                r = op2(VM_DIV, diff(a, k, dcache, dep), op2(VM_MUL, op1(VM_COS, a), op1(VM_COS, a)));
                break;
            case VM_ATAN:
                r = op2(VM_DIV, diff(a, k, dcache, dep), op2(VM_ADD, cst(1.0), op2(VM_MUL, a, a)));
                break;
            case VM_SQRT:
                r = op2(VM_DIV, diff(a, k, dcache, dep), op2(VM_MUL, cst(2.0), n));
                break;
            case VM_ABS:
                r = op2(VM_MUL, op1(VM_SIGN, a), diff(a, k, dcache, dep));
                break;
            case VM_TANH:
                r = op2(VM_MUL, op2(VM_SUB, cst(1.0), op2(VM_MUL, n, n)), diff(a, k, dcache, dep));
                break;
            case VM_SINH:
                r = op2(VM_MUL, op1(VM_COSH, a), diff(a, k, dcache, dep));
                break;
            case VM_COSH:
                r = op2(VM_MUL, op1(VM_SINH, a), diff(a, k, dcache, dep));
                break;
            case VM_ASIN: // 1 / sqrt(1 - a^2)
                r = op2(VM_DIV, diff(a, k, dcache, dep), op1(VM_SQRT, op2(VM_SUB, cst(1.0), op2(VM_MUL, a, a))));
                break;
            case VM_ACOS:
                r = op1(VM_NEG, op2(VM_DIV, diff(a, k, dcache, dep), op1(VM_SQRT, op2(VM_SUB, cst(1.0), op2(VM_MUL, a, a)))));
                break;
            case VM_LOG1P:
                r = op2(VM_DIV, diff(a, k, dcache, dep), op2(VM_ADD, cst(1.0), a));
                break;
            case VM_EXPM1:
                r = op2(VM_MUL, op1(VM_EXP, a), diff(a, k, dcache, dep));
                break;
            case VM_GAMMA: // gamma(a) digamma(a) a'
                r = op2(VM_MUL, op2(VM_MUL, n, op1(VM_PSI0, a)), diff(a, k, dcache, dep));
                break;
            case VM_LGAMMA:
                r = op2(VM_MUL, op1(VM_PSI0, a), diff(a, k, dcache, dep));
                break;
            case VM_PSI0:
            case VM_PSI1:
            case VM_PSI2:
            case VM_PSI3: // psigamma(a, n)' = psigamma(a, n + 1) a'
                r = op2(VM_MUL, op1((unsigned char)(nd.op + 1), a), diff(a, k, dcache, dep));
                break;
            case VM_PSI4:
                error = "psigamma: derivative beyond order 4";
                r = cst(NAN);
                break;
            case VM_PNORM: // dnorm(a) a'
                r = op2(VM_MUL, op2(VM_MUL, op1(VM_EXP, op2(VM_MUL, cst(-0.5), op2(VM_MUL, a, a))), cst(0.39894228040143267794)),
                        diff(a, k, dcache, dep));
                break;
            default:
                r = cst(NAN);
            }
        }
        dcache[n] = r;
        return r;
    }

    // emit: topological order, the value's closure first
    // fvv < 0: no third closure
    template <class Prog>
    bool emit(int value, const std::vector<int> &grads, int fvv, int p, int nx, Prog &out)
    {
        memset(&out, 0, sizeof(out));
        out.p = p;
        out.nx = nx;
        std::map<int, int> slot_of; // node -> slot
        std::vector<double> consts;
        std::vector<int> order;     // op nodes in emission order
        std::vector<char> seen(nodes.size(), 0);
        // constants first (collect), then ops by DFS
        std::vector<int> stack;
        auto visit = [&](int root) {
            stack.push_back(root);
            std::vector<std::pair<int, int>> st; // (node, state)
            st.push_back({root, 0});
            while (!st.empty())
            {
                auto &top = st.back();
                const int n = top.first;
                if (seen[n])
                {
                    st.pop_back();
                    continue;
                }
                const Node &nd = nodes[n];
                if (nd.kind != K_OP)
                {
                    seen[n] = 1;
                    if (nd.kind == K_CONST)
                        consts.push_back(nd.c), slot_of[n] = -(int)consts.size(); // placeholder, fixed below
                    st.pop_back();
                    continue;
                }
                if (top.second == 0)
                {
                    top.second = 1;
                    if (!seen[nd.a])
                        st.push_back({nd.a, 0});
                    if (nd.b != nd.a && !seen[nd.b])
                        st.push_back({nd.b, 0});
                }
                else
                {
                    seen[n] = 1;
                    order.push_back(n);
                    st.pop_back();
                }
            }
        };
        visit(value);
        const int nvalue = (int)order.size();
        for (int g : grads)
            visit(g);
        const int ngrad = (int)order.size();
        if (fvv >= 0)
            visit(fvv);
        if ((int)consts.size() > Prog::CAP_CONST || (int)order.size() > Prog::CAP_OPS ||
            2 * p + nx + (int)consts.size() + (int)order.size() > 65535)
        {
            error = "expression too large for the device program";
            return false;
        }
        out.nconst = (int)consts.size();
        for (size_t c = 0; c < consts.size(); ++c)
            out.consts[c] = consts[c];
        const int base = 2 * p + nx + out.nconst;
        auto slot = [&](int n) -> int {
            const Node &nd = nodes[n];
            if (nd.kind == K_PARAM)
                return nd.a;
            if (nd.kind == K_VAR)
                return p + nd.a;
            if (nd.kind == K_VDIR)
                return p + nx + nd.a;
            if (nd.kind == K_CONST)
                return 2 * p + nx + (-slot_of[n] - 1);
            return slot_of[n];
        };
        for (size_t i = 0; i < order.size(); ++i)
        {
            const Node &nd = nodes[order[i]];
            out.op[i] = nd.op;
            out.a[i] = (unsigned short)slot(nd.a);
            out.b[i] = (unsigned short)slot(nd.b);
            // (the interpreter always reads both operands: a unary instruction names its operand twice)
            if constexpr (sizeof(out.word) / sizeof(out.word[0]) > 1)
            {
                const unsigned int wb = vm_is_binary(out.op[i]) ? out.b[i] : out.a[i];
                out.word[i] = (unsigned int)out.op[i] | ((unsigned int)out.a[i] << 8) | (wb << 20);
            }
            slot_of[order[i]] = base + (int)i;
        }
        out.nops = ngrad;
        out.nfvv = fvv >= 0 ? (int)order.size() : 0;
        out.fvv_slot = fvv >= 0 ? slot(fvv) : 0;
        out.nvalue = nvalue;
        out.value_slot = slot(value);
        for (int k = 0; k < p; ++k)
            out.grad_slot[k] = slot(grads[k]);
        return true;
    }
};

// rhs text + names -> program.  Returns "" on success or an error message.  nx_slots: regressor slots of the program's
// slot layout (the interpreter's kernels always carry VM_NX columns; the wide path as many as the formula names)
template <class Prog>
inline std::string compile_expression_t(const char *rhs, const std::vector<std::string> &parnames,
                                        const std::vector<std::string> &varnames, int nx_slots, Prog &out)
{
    if ((int)parnames.size() > Prog::CAP_P)
        return "too many parameters for an expression model";
    if ((int)varnames.size() > nx_slots)
        return "too many regressors for an expression model";
    FParser fp(rhs);
    FNodeP ast = fp.parse();
    if (!ast)
        return "cannot parse expression";
    ExprCompiler ec;
    const int value = ec.build(ast, parnames, varnames);
    if (!ec.error.empty())
        return ec.error;
    std::vector<int> grads;
    for (size_t k = 0; k < parnames.size(); ++k)
    {
        std::map<int, int> dcache;
        std::map<int, bool> dep;
        grads.push_back(ec.diff(value, (int)k, dcache, dep));
    }
    // second directional derivative D^2 f[v, v] = sum_j v_j d/dtheta_j ( sum_k v_k df/dtheta_k )
    const int p = (int)parnames.size();
    int g = ec.cst(0.0);
    for (int k = 0; k < p; ++k)
        g = ec.op2(VM_ADD, g, ec.op2(VM_MUL, ec.vdir(k), grads[k]));
    int fvv = ec.cst(0.0);
    for (int j = 0; j < p; ++j)
    {
        std::map<int, int> dcache;
        std::map<int, bool> dep;
        fvv = ec.op2(VM_ADD, fvv, ec.op2(VM_MUL, ec.vdir(j), ec.diff(g, j, dcache, dep)));
    }
    if (ec.emit(value, grads, fvv, p, nx_slots, out))
        return "";
    // too large with the third closure: keep value + gradient (fvv then falls back to finite differences)
    ec.error.clear();
    if (!ec.emit(value, grads, -1, p, nx_slots, out))
        return ec.error;
    return "";
}

inline std::string compile_expression(const char *rhs, const std::vector<std::string> &parnames,
                                      const std::vector<std::string> &varnames, VmProgram &out)
{
    return compile_expression_t(rhs, parnames, varnames, VM_NX, out);
}

} // namespace gslnls
