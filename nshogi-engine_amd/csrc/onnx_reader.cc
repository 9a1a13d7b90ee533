// onnx_reader.cc -- see onnx_reader.h.
#include "onnx_reader.h"

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>

namespace nsg {
namespace onnx {

namespace {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- protobuf wire format -------------------------------------------------------------------
struct Span {
    const unsigned char* p = nullptr;
    size_t n = 0;
};

struct Field {
    uint32_t number = 0;
    int wire = 0;
    uint64_t value = 0; // varint / fixed
    Span bytes;         // length-delimited, fixed32, fixed64
};

class Reader {
 public:
    explicit Reader(Span S) : P(S.p), End(S.p + S.n) {}
    bool next(Field* F) {
        if (P >= End) return false;
        const uint64_t Key = varint();
        F->number = (uint32_t)(Key >> 3);
        F->wire = (int)(Key & 7);
        F->value = 0;
        F->bytes = Span{}; // a varint field must not leave the previous field's bytes behind
        switch (F->wire) {
        case 0: F->value = varint(); break;
        case 1: F->bytes = take(8); break;
        case 2: F->bytes = take((size_t)varint()); break;
        case 5: F->bytes = take(4); break;
        default: throw Error("unsupported protobuf wire type " + std::to_string(F->wire));
        }
        return true;
    }
    uint64_t varint() {
        uint64_t V = 0;
        for (int Shift = 0; Shift < 70; Shift += 7) {
            if (P >= End) throw Error("truncated protobuf varint");
            const unsigned char B = *P++;
            V |= (uint64_t)(B & 0x7F) << Shift;
            if (!(B & 0x80)) return V;
        }
        throw Error("malformed protobuf varint");
    }
    bool done() const { return P >= End; }

 private:
    Span take(size_t N) {
        if ((size_t)(End - P) < N) throw Error("truncated protobuf field");
        Span S{P, N};
        P += N;
        return S;
    }
    const unsigned char* P;
    const unsigned char* End;
};

std::string str(Span S) { return std::string((const char*)S.p, S.n); }

float f32(Span S) {
    if (!S.p || S.n < 4) throw Error("malformed protobuf: a float field is not 4 bytes wide");
    float V;
    std::memcpy(&V, S.p, 4);
    return V;
}

// a field that must be length-delimited (strings, sub-messages, raw_data)
Span bytesOf(const Field& F, const char* What) {
    if (F.wire != 2) throw Error(std::string("malformed protobuf: ") + What + " is not length-delimited");
    return F.bytes;
}

// ---- ONNX messages (onnx.proto3 field numbers) ----------------------------------------------
struct Tensor {
    std::vector<int64_t> Dims;
    std::vector<float> F;   // data_type FLOAT (1)
    std::vector<int64_t> I; // data_type INT64 (7)
    bool IsFloat = true;
    size_t count() const { return IsFloat ? F.size() : I.size(); }
};

Tensor readTensor(Span S, std::string* Name) {
    Tensor T;
    int DataType = 0;
    Span Raw;
    bool HasRaw = false;
    Reader R(S);
    Field Fd;
    while (R.next(&Fd)) {
        switch (Fd.number) {
        case 1: // dims
            if (Fd.wire == 0) T.Dims.push_back((int64_t)Fd.value);
            else { Reader P(Fd.bytes); while (!P.done()) T.Dims.push_back((int64_t)P.varint()); }
            break;
        case 2: DataType = (int)Fd.value; break;
        case 4: // float_data
            if (Fd.wire == 5) T.F.push_back(f32(Fd.bytes));
            else if (Fd.wire == 2) for (size_t K = 0; K + 4 <= Fd.bytes.n; K += 4) T.F.push_back(f32(Span{Fd.bytes.p + K, 4}));
            else throw Error("malformed protobuf: float_data is neither fixed32 nor packed");
            break;
        case 7: // int64_data
            if (Fd.wire == 0) T.I.push_back((int64_t)Fd.value);
            else { Reader P(Fd.bytes); while (!P.done()) T.I.push_back((int64_t)P.varint()); }
            break;
        case 8: if (Name) *Name = str(bytesOf(Fd, "a tensor name")); break;
        case 9: Raw = bytesOf(Fd, "raw_data"); HasRaw = true; break;
        case 14: if (Fd.value != 0) throw Error("initializer with external data: not supported (keep the weights inside the .onnx file)"); break;
        default: break;
        }
    }
    if (DataType == 1) {
        T.IsFloat = true;
        if (HasRaw) {
            T.F.resize(Raw.n / 4);
            std::memcpy(T.F.data(), Raw.p, T.F.size() * 4);
        }
    } else if (DataType == 7) {
        T.IsFloat = false;
        if (HasRaw) {
            T.I.resize(Raw.n / 8);
            std::memcpy(T.I.data(), Raw.p, T.I.size() * 8);
        }
    } else {
        throw Error("initializer '" + (Name ? *Name : std::string()) + "': unsupported data type " +
                    std::to_string(DataType) + " (float32 / int64 only)");
    }
    size_t Want = 1;
    for (int64_t D : T.Dims) Want *= (size_t)D;
    if (Want != T.count()) throw Error("initializer '" + (Name ? *Name : std::string()) + "': element count does not match its dims");
    return T;
}

struct Attr {
    bool HasF = false, HasI = false;
    float F = 0.f;
    int64_t I = 0;
    std::vector<int64_t> Ints;
    std::shared_ptr<Tensor> T;
};

struct Node {
    std::string Op, Name;
    std::vector<std::string> In, Out;
    std::map<std::string, Attr> Attrs;
    int64_t attrI(const char* K, int64_t Default) const {
        auto It = Attrs.find(K);
        return It != Attrs.end() && It->second.HasI ? It->second.I : Default;
    }
    double attrF(const char* K, double Default) const {
        auto It = Attrs.find(K);
        return It != Attrs.end() && It->second.HasF ? (double)It->second.F : Default;
    }
    std::vector<int64_t> attrInts(const char* K, std::vector<int64_t> Default) const {
        auto It = Attrs.find(K);
        return It != Attrs.end() && !It->second.Ints.empty() ? It->second.Ints : Default;
    }
};

Node readNode(Span S) {
    Node N;
    Reader R(S);
    Field Fd;
    while (R.next(&Fd)) {
        switch (Fd.number) {
        case 1: N.In.push_back(str(bytesOf(Fd, "a node input"))); break;
        case 2: N.Out.push_back(str(bytesOf(Fd, "a node output"))); break;
        case 3: N.Name = str(bytesOf(Fd, "a node name")); break;
        case 4: N.Op = str(bytesOf(Fd, "an op type")); break;
        case 5: {
            std::string Name;
            Attr A;
            Reader AR(bytesOf(Fd, "an attribute"));
            Field Af;
            while (AR.next(&Af)) {
                switch (Af.number) {
                case 1: Name = str(bytesOf(Af, "an attribute name")); break;
                case 2:
                    if (Af.wire != 5) throw Error("malformed protobuf: attribute float is not fixed32");
                    A.F = f32(Af.bytes); A.HasF = true; break;
                case 3: A.I = (int64_t)Af.value; A.HasI = true; break;
                case 5: A.T = std::make_shared<Tensor>(readTensor(bytesOf(Af, "an attribute tensor"), nullptr)); break;
                case 8:
                    if (Af.wire == 0) A.Ints.push_back((int64_t)Af.value);
                    else { Reader P(Af.bytes); while (!P.done()) A.Ints.push_back((int64_t)P.varint()); }
                    break;
                default: break;
                }
            }
            N.Attrs[Name] = A;
            break;
        }
        default: break;
        }
    }
    return N;
}

std::string valueInfoName(Span S) {
    Reader R(S);
    Field Fd;
    while (R.next(&Fd))
        if (Fd.number == 1) return str(bytesOf(Fd, "a value-info name"));
    return std::string();
}

struct Graph {
    std::vector<Node> Nodes;
    std::map<std::string, Tensor> Inits;
    std::vector<std::string> Inputs, Outputs;
    std::map<std::string, std::vector<const Node*>> Consumers;
};

Graph readModel(Span Data) {
    Span GraphBytes;
    bool HaveGraph = false;
    {
        Reader R(Data);
        Field Fd;
        while (R.next(&Fd))
            if (Fd.number == 7 && Fd.wire == 2) { GraphBytes = Fd.bytes; HaveGraph = true; }
    }
    if (!HaveGraph) throw Error("not an ONNX ModelProto: no graph");
    Graph G;
    Reader R(GraphBytes);
    Field Fd;
    while (R.next(&Fd)) {
        if (Fd.wire != 2) continue;
        if (Fd.number == 1) G.Nodes.push_back(readNode(Fd.bytes));
        else if (Fd.number == 5) { std::string Name; Tensor T = readTensor(Fd.bytes, &Name); G.Inits[Name] = std::move(T); }
        else if (Fd.number == 11) G.Inputs.push_back(valueInfoName(Fd.bytes));
        else if (Fd.number == 12) G.Outputs.push_back(valueInfoName(Fd.bytes));
    }
    // Constant nodes are initializers in all but name (torch emits them for scalar literals)
    for (const Node& N : G.Nodes) {
        if (N.Op != "Constant" || N.Out.size() != 1) continue;
        auto It = N.Attrs.find("value");
        if (It != N.Attrs.end() && It->second.T) G.Inits[N.Out[0]] = *It->second.T;
        else if ((It = N.Attrs.find("value_float")) != N.Attrs.end() && It->second.HasF) {
            Tensor T;
            T.F.push_back(It->second.F);
            G.Inits[N.Out[0]] = T;
        }
    }
    for (const Node& N : G.Nodes)
        for (const std::string& I : N.In) G.Consumers[I].push_back(&N);
    return G;
}

// ---- topology matching ----------------------------------------------------------------------
[[noreturn]] void fail(const Node& N, const std::string& Why) {
    throw Error("unsupported ONNX structure at node '" + N.Name + "' (" + N.Op + "): " + Why);
}

struct ConvBn {
    std::vector<float> W;     // [n][cin][k][k]
    std::vector<float> Stats; // [4][n]: gamma, beta, mean, var
    int N = 0, Cin = 0;
    bool NoBn = false; // identity statistics: var is set to 1 - eps once the model's epsilon is known
    std::string Out;
};

class Matcher {
 public:
    explicit Matcher(const Graph& Gr) : G(Gr) {}

    const Node& only(const std::string& T, const char* What) const {
        auto It = G.Consumers.find(T);
        const size_t C = It == G.Consumers.end() ? 0 : It->second.size();
        if (C != 1) throw Error(std::string(What) + ": tensor '" + T + "' has " + std::to_string(C) + " consumers, expected 1");
        return *It->second[0];
    }
    std::vector<const Node*> consumers(const std::string& T) const {
        auto It = G.Consumers.find(T);
        return It == G.Consumers.end() ? std::vector<const Node*>() : It->second;
    }
    const Tensor* init(const std::string& Name) const {
        auto It = G.Inits.find(Name);
        return It == G.Inits.end() ? nullptr : &It->second;
    }
    const Tensor& needFloat(const Node& N, size_t Idx, const char* What) const {
        const Tensor* T = Idx < N.In.size() ? init(N.In[Idx]) : nullptr;
        if (!T || !T->IsFloat) fail(N, std::string("expected a constant float32 ") + What);
        return *T;
    }

    ConvBn convBn(const std::string& X, const Node& Nd, int K) {
        if (Nd.Op != "Conv" || Nd.In.empty() || Nd.In[0] != X) fail(Nd, "expected a Conv of '" + X + "'");
        const Tensor& Wt = needFloat(Nd, 1, "weight");
        if (Wt.Dims.size() != 4 || Wt.Dims[2] != K || Wt.Dims[3] != K)
            fail(Nd, "expected a constant " + std::to_string(K) + "x" + std::to_string(K) + " weight");
        const int64_t Pad = K / 2;
        bool Ok = Nd.attrI("group", 1) == 1;
        for (int64_t S : Nd.attrInts("strides", {1, 1})) Ok = Ok && S == 1;
        for (int64_t D : Nd.attrInts("dilations", {1, 1})) Ok = Ok && D == 1;
        const std::vector<int64_t> Pads = Nd.attrInts("pads", {Pad, Pad, Pad, Pad});
        Ok = Ok && Pads.size() == 4;
        for (int64_t P : Pads) Ok = Ok && P == Pad;
        if (!Ok) fail(Nd, "only stride 1, dilation 1, group 1, 'same' padding");
        ConvBn R;
        R.N = (int)Wt.Dims[0];
        R.Cin = (int)Wt.Dims[1];
        R.W = Wt.F;
        std::vector<float> Bias((size_t)R.N, 0.f);
        if (Nd.In.size() > 2 && !Nd.In[2].empty()) {
            const Tensor& B = needFloat(Nd, 2, "bias");
            if ((int)B.count() != R.N) fail(Nd, "bias length does not match the output channels");
            Bias = B.F;
        }
        R.Stats.resize((size_t)4 * R.N);
        const std::string& Y = Nd.Out[0];
        const auto Next = consumers(Y);
        if (Next.size() == 1 && Next[0]->Op == "BatchNormalization") {
            const Node& B = *Next[0];
            if (B.In.size() < 5) fail(B, "expected scale, bias, mean and variance inputs");
            for (int S = 0; S < 4; ++S) {
                const Tensor& T = needFloat(B, (size_t)S + 1, "BatchNormalization statistic");
                if ((int)T.count() != R.N) fail(B, "statistic length does not match the channels");
                for (int I = 0; I < R.N; ++I) R.Stats[(size_t)S * R.N + I] = T.F[I];
            }
            for (int I = 0; I < R.N; ++I) R.Stats[(size_t)2 * R.N + I] -= Bias[I]; // a conv bias folds into the BN mean
            EpsSeen.push_back(B.attrF("epsilon", kEpsDefault));
            R.Out = B.Out[0];
            return R;
        }
        // no BN: identity statistics carrying the conv bias (var + eps == 1: folded scale 1; the
        // variance is rewritten by finishIdentity() with the epsilon the NSGW header will hold)
        R.NoBn = true;
        for (int I = 0; I < R.N; ++I) {
            R.Stats[I] = 1.f;
            R.Stats[(size_t)R.N + I] = Bias[I];
            R.Stats[(size_t)2 * R.N + I] = 0.f;
            R.Stats[(size_t)3 * R.N + I] = (float)(1.0 - kEpsDefault);
        }
        R.Out = Y;
        return R;
    }

    std::string relu(const std::string& T) const {
        const Node& Nd = only(T, "activation");
        if (Nd.Op != "Relu") fail(Nd, "expected Relu");
        return Nd.Out[0];
    }

    bool reaches(const std::string& From, const std::string& Goal) const {
        std::set<std::string> Seen;
        std::vector<std::string> Todo{From};
        while (!Todo.empty()) {
            const std::string U = Todo.back();
            Todo.pop_back();
            if (U == Goal) return true;
            for (const Node* C : consumers(U))
                for (const std::string& O : C->Out)
                    if (Seen.insert(O).second) Todo.push_back(O);
        }
        return false;
    }

    struct Dense {
        std::vector<float> W; // [out][in]
        std::vector<float> B;
        int Out = 0, In = 0;
        std::string Y;
    };

    Dense dense(const std::string& T, const Node& Nd) const {
        Dense D;
        if (Nd.Op == "Gemm" && !Nd.In.empty() && Nd.In[0] == T) {
            if (Nd.attrF("alpha", 1.0) != 1.0 || Nd.attrF("beta", 1.0) != 1.0 || Nd.attrI("transA", 0) != 0)
                fail(Nd, "Gemm with alpha = beta = 1, transA = 0 only");
            const Tensor& Wt = needFloat(Nd, 1, "weight");
            if (Wt.Dims.size() != 2) fail(Nd, "expected a 2-D weight");
            const bool TransB = Nd.attrI("transB", 0) != 0;
            D.Out = (int)(TransB ? Wt.Dims[0] : Wt.Dims[1]);
            D.In = (int)(TransB ? Wt.Dims[1] : Wt.Dims[0]);
            D.W.resize((size_t)D.Out * D.In);
            for (int O = 0; O < D.Out; ++O)
                for (int I = 0; I < D.In; ++I)
                    D.W[(size_t)O * D.In + I] = TransB ? Wt.F[(size_t)O * D.In + I] : Wt.F[(size_t)I * D.Out + O];
            D.B.assign((size_t)D.Out, 0.f);
            if (Nd.In.size() > 2 && !Nd.In[2].empty()) {
                const Tensor& B = needFloat(Nd, 2, "bias");
                if ((int)B.count() != D.Out) fail(Nd, "bias length does not match");
                D.B = B.F;
            }
            D.Y = Nd.Out[0];
            return D;
        }
        if (Nd.Op == "MatMul" && !Nd.In.empty() && Nd.In[0] == T) {
            const Tensor& Wt = needFloat(Nd, 1, "weight");
            if (Wt.Dims.size() != 2) fail(Nd, "expected a 2-D weight");
            D.In = (int)Wt.Dims[0];
            D.Out = (int)Wt.Dims[1];
            D.W.resize((size_t)D.Out * D.In);
            for (int O = 0; O < D.Out; ++O)
                for (int I = 0; I < D.In; ++I) D.W[(size_t)O * D.In + I] = Wt.F[(size_t)I * D.Out + O];
            const Node& Add = only(Nd.Out[0], "bias add");
            if (Add.Op != "Add") fail(Add, "expected MatMul + Add");
            const Tensor* B = nullptr;
            for (const std::string& I : Add.In)
                if (const Tensor* C = init(I)) B = C;
            if (!B || !B->IsFloat || (int)B->count() != D.Out) fail(Add, "expected a constant bias");
            D.B = B->F;
            D.Y = Add.Out[0];
            return D;
        }
        fail(Nd, "expected Gemm or MatMul+Add");
    }

    // the scalar constant among a binary node's inputs (the other input must be T)
    double scalarOperand(const Node& Nd, const std::string& T) const {
        if (Nd.In.size() != 2) fail(Nd, "expected two inputs");
        const bool First = Nd.In[0] == T;
        if (!First && Nd.In[1] != T) fail(Nd, "unexpected operand");
        if (Nd.Op == "Div" && !First) fail(Nd, "constant / x is not part of the value head");
        const Tensor* C = init(Nd.In[First ? 1 : 0]);
        if (!C || !C->IsFloat || C->count() != 1) fail(Nd, "expected a scalar float constant");
        return (double)C->F[0];
    }

    static constexpr double kEpsDefault = 1e-5;
    std::vector<double> EpsSeen;

 private:
    const Graph& G;
};

void put(std::vector<unsigned char>* Blob, const void* P, size_t N) {
    const unsigned char* B = (const unsigned char*)P;
    Blob->insert(Blob->end(), B, B + N);
}

void convert(Span Data, std::vector<unsigned char>* Blob) {
    const Graph G = readModel(Data);
    std::set<std::string> Ins;
    for (const std::string& I : G.Inputs)
        if (!G.Inits.count(I)) Ins.insert(I);
    const std::set<std::string> Outs(G.Outputs.begin(), G.Outputs.end());
    if (!Ins.count("input") || !Outs.count("policy") || !Outs.count("value") || !Outs.count("draw"))
        throw Error("tensor contract (trt.cc:144-227): need input 'input' and outputs 'policy', 'value', 'draw'");
    Matcher M(G);

    ConvBn Stem = M.convBn("input", M.only("input", "stem"), 3);
    std::string X = M.relu(Stem.Out);
    std::vector<ConvBn> Blocks; // w1, w2, w1, w2, ...
    for (;;) {
        const auto Cs = M.consumers(X);
        const Node* Conv3 = nullptr;
        const Node* Add = nullptr;
        int NumConv3 = 0, NumAdd = 0;
        for (const Node* C : Cs) {
            if (C->Op == "Add") { Add = C; ++NumAdd; }
            if (C->Op == "Conv" && C->In.size() > 1) {
                const Tensor* W = M.init(C->In[1]);
                if (W && W->Dims.size() == 4 && W->Dims[2] == 3 && W->Dims[3] == 3) { Conv3 = C; ++NumConv3; }
            }
        }
        if (!(NumConv3 == 1 && NumAdd == 1 && Cs.size() == 2)) break;
        ConvBn A = M.convBn(X, *Conv3, 3);
        const std::string Y = M.relu(A.Out);
        ConvBn B = M.convBn(Y, M.only(Y, "second conv of a block"), 3);
        const Node& Sum = M.only(B.Out, "residual add");
        const bool Operands = Sum.In.size() == 2 && ((Sum.In[0] == X && Sum.In[1] == B.Out) || (Sum.In[1] == X && Sum.In[0] == B.Out));
        if (&Sum != Add || !Operands) fail(Sum, "expected x + conv path");
        X = M.relu(Sum.Out[0]);
        Blocks.push_back(std::move(A));
        Blocks.push_back(std::move(B));
    }
    const auto Heads = M.consumers(X);
    if (Heads.size() != 2 || Heads[0]->Op != "Conv" || Heads[1]->Op != "Conv")
        throw Error("after " + std::to_string(Blocks.size() / 2) + " residual blocks expected the policy and value 1x1 convs");
    const bool FirstIsPolicy = M.reaches(Heads[0]->Out[0], "policy");
    const bool SecondIsPolicy = M.reaches(Heads[1]->Out[0], "policy");
    if (FirstIsPolicy == SecondIsPolicy) throw Error("cannot tell the policy head from the value head");
    const Node& PolNode = *Heads[FirstIsPolicy ? 0 : 1];
    const Node& ValNode = *Heads[FirstIsPolicy ? 1 : 0];

    ConvBn Pol = M.convBn(X, PolNode, 1);
    for (int I = 0; I < Pol.N; ++I)
        if (!Pol.NoBn)
            fail(PolNode, "the policy conv must be followed directly by the flatten (no BN)");
    {
        const Node& Fl = M.only(Pol.Out, "policy flatten");
        if ((Fl.Op != "Flatten" && Fl.Op != "Reshape") || Fl.Out[0] != "policy") fail(Fl, "expected Flatten/Reshape -> policy");
    }
    std::vector<float> PolB((size_t)Pol.N);
    for (int I = 0; I < Pol.N; ++I) PolB[I] = Pol.Stats[(size_t)Pol.N + I] - Pol.Stats[(size_t)2 * Pol.N + I];

    ConvBn Val = M.convBn(X, ValNode, 1);
    const std::string VRelu = M.relu(Val.Out);
    const Node& VFl = M.only(VRelu, "value flatten");
    if (VFl.Op != "Flatten" && VFl.Op != "Reshape") fail(VFl, "expected Flatten/Reshape");
    Matcher::Dense Fc1 = M.dense(VFl.Out[0], M.only(VFl.Out[0], "value MLP layer 1"));
    const std::string H = M.relu(Fc1.Y);

    std::vector<float> Fc2W((size_t)2 * Fc1.Out, 0.f);
    float Fc2B[2] = {0.f, 0.f};
    bool Have[2] = {false, false};
    for (const Node* Nd : M.consumers(H)) {
        Matcher::Dense D = M.dense(H, *Nd);
        if (D.Out != 1) fail(*Nd, "one dense node per output expected (value, draw)");
        float Scale = 1.f;
        std::string Name;
        const Node& Sq = M.only(D.Y, "output squashing");
        if (Sq.Op == "Tanh") { // (tanh(o) + 1) / 2
            const Node& A = M.only(Sq.Out[0], "tanh shift");
            if (A.Op != "Add" || M.scalarOperand(A, Sq.Out[0]) != 1.0) fail(A, "expected tanh + 1");
            const Node& S = M.only(A.Out[0], "tanh scale");
            const bool Half = (S.Op == "Mul" && M.scalarOperand(S, A.Out[0]) == 0.5) ||
                              (S.Op == "Div" && M.scalarOperand(S, A.Out[0]) == 2.0);
            if (!Half) fail(S, "expected (tanh + 1) * 0.5 or (tanh + 1) / 2");
            Name = S.Out[0];
            if (Name == "draw") Scale = 2.f; // the device evaluates draw as sigmoid(o): (tanh(z) + 1) / 2 = sigmoid(2 z)
        } else if (Sq.Op == "Sigmoid") {
            Name = Sq.Out[0];
            if (Name == "value") Scale = 0.5f; // sigmoid(z) = (tanh(z / 2) + 1) / 2
        } else {
            fail(Sq, "expected Tanh or Sigmoid");
        }
        const int Row = Name == "value" ? 0 : Name == "draw" ? 1 : -1;
        if (Row < 0) fail(Sq, "the value MLP must end in the outputs 'value' and 'draw'");
        if (D.In != Fc1.Out) fail(*Nd, "width does not match the hidden layer");
        for (int I = 0; I < D.In; ++I) Fc2W[(size_t)Row * Fc1.Out + I] = D.W[I] * Scale;
        Fc2B[Row] = D.B[0] * Scale;
        Have[Row] = true;
    }
    if (!Have[0] || !Have[1]) throw Error("value MLP outputs: need both 'value' and 'draw'");

    const int F = Stem.N, Cin = Stem.Cin, NumBlocks = (int)Blocks.size() / 2;
    for (const ConvBn& B : Blocks)
        if (B.N != F || B.Cin != F) throw Error("residual blocks must keep the trunk width");
    if (Pol.Cin != F || Val.Cin != F) throw Error("head convolutions must read the trunk");
    if (Fc1.In != Val.N * 81) throw Error("value MLP layer 1 must read value_channels * 81 features");
    if (Pol.N * 81 != 2187) throw Error("Unexpected PolicySize: " + std::to_string(Pol.N * 81) + " (expected: 2187)."); // trt.cc:205
    double Eps = M.EpsSeen.empty() ? Matcher::kEpsDefault : M.EpsSeen[0];
    for (double E : M.EpsSeen)
        if (std::fabs(E - Eps) > 1e-12) throw Error("the BatchNormalization nodes use different epsilons; the NSGW header holds one");
    // a conv without BN folds with scale 1 only if its variance + THIS model's epsilon is 1
    auto FinishIdentity = [&](ConvBn* C) {
        if (!C->NoBn) return;
        for (int I = 0; I < C->N; ++I) C->Stats[(size_t)3 * C->N + I] = (float)(1.0 - (double)(float)Eps);
    };
    FinishIdentity(&Stem);
    for (ConvBn& B : Blocks) FinishIdentity(&B);
    FinishIdentity(&Val);

    Blob->clear();
    unsigned char Header[64] = {0};
    std::memcpy(Header, "NSGW", 4);
    const uint32_t Hd[7] = {1u, (uint32_t)Cin, (uint32_t)F, (uint32_t)NumBlocks, (uint32_t)Pol.N, (uint32_t)Val.N, (uint32_t)Fc1.Out};
    std::memcpy(Header + 4, Hd, sizeof(Hd));
    const float EpsF = (float)Eps;
    std::memcpy(Header + 32, &EpsF, 4);
    put(Blob, Header, 64);
    auto PutF = [&](const std::vector<float>& V) { put(Blob, V.data(), V.size() * 4); };
    PutF(Stem.W); PutF(Stem.Stats);
    for (const ConvBn& B : Blocks) { PutF(B.W); PutF(B.Stats); }
    PutF(Pol.W); PutF(PolB);
    PutF(Val.W); PutF(Val.Stats);
    PutF(Fc1.W); PutF(Fc1.B);
    PutF(Fc2W);
    put(Blob, Fc2B, 8);
}

} // namespace

bool isNsgw(const void* Data, size_t Size) {
    return Size >= 4 && std::memcmp(Data, "NSGW", 4) == 0;
}

bool convertToNsgw(const void* Data, size_t Size, std::vector<unsigned char>* Blob, std::string* Error_) {
    try {
        convert(Span{(const unsigned char*)Data, Size}, Blob);
        return true;
    } catch (const std::exception& E) {
        if (Error_) *Error_ = E.what();
        return false;
    }
}

} // namespace onnx
} // namespace nsg
