# FastConvHIP.jl -- Julia binding of liblsfc.so (include/lsfc.h) that keeps the reference's
# operator surface, so `gmres!(u, fastconv, rhs, Pl=precond)` and `LinearMap(...)` run unchanged
# (examples/example.jl:54-93, examples/example3D.jl:54-79 of the reference).
#
# UNVERIFIED: Julia is not installed in the build image; this file is the binding a maintainer
# would add, kept declarative over the C ABI.  The tested host side is the Python mirror.
module FastConvHIP

using LinearAlgebra
import Base: *, size, eltype

const liblsfc = get(ENV, "LSFC_LIB", joinpath(@__DIR__, "..", "fast_solver_lippmann_schwinger_amd", "liblsfc.so"))
const QUAD = Dict("trapezoidal" => Cint(0), "Greengard_Vico" => Cint(1))

lasterror() = unsafe_string(ccall((:lsfc_last_error, liblsfc), Cstring, ()))
check(rc) = rc == 0 ? nothing : error("lsfc error $rc: $(lasterror())")

mutable struct FastMHIP            # mirrors FastM / FastM3D field names (src/FastConvolution.jl:11-27)
    plan::Ptr{Cvoid}
    nu::Vector{Float64}
    n::Int64; m::Int64; l::Int64
    omega::Float64
    quadRule::String
    function FastMHIP(plan, nu, n, m, l, omega, quadRule)
        M = new(plan, nu, n, m, l, omega, quadRule)
        finalizer(M -> ccall((:lsfc_plan_destroy, liblsfc), Cint, (Ptr{Cvoid},), M.plan), M)
        return M
    end
end

# FastM(GFFT,nu,ne,me,n,m,k; quadRule) -- src/FastConvolution.jl:24
function FastM(GFFT::Array{Complex{Float64},2}, nu::Vector{Float64}, ne, me, n, m, k; quadRule::String="trapezoidal", flags=0, device=0)
    plan = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lsfc_plan_create_2d, liblsfc), Cint,
                (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Complex{Float64}}, Float64, Cint, Cuint, Cint),
                plan, n, m, ne, me, nu, GFFT, k, QUAD[quadRule], flags, device))
    FastMHIP(plan[], nu, n, m, 1, k, quadRule)
end

# FastM3D(GFFT,nu,ne,me,le,n,m,l,k; quadRule) -- src/FastConvolution3D.jl:23
function FastM3D(GFFT::Array{Complex{Float64},3}, nu::Vector{Float64}, ne, me, le, n, m, l, k; quadRule::String="Greengard_Vico", flags=0, device=0)
    plan = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lsfc_plan_create_3d, liblsfc), Cint,
                (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Complex{Float64}}, Float64, Cint, Cuint, Cint),
                plan, n, m, l, ne, me, le, nu, GFFT, k, QUAD[quadRule], flags, device))
    FastMHIP(plan[], nu, n, m, l, k, quadRule)
end

# buildFastConvolution3D(x,y,z,X,Y,Z,h,k,nu) -- src/FastConvolution3D.jl:68 (symbol generated on the device)
function buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nu; quadRule::String="Greengard_Vico", flags=0, device=0)
    nuv = Vector{Float64}(nu(X, Y, Z)); plan = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lsfc_plan_create_gv3d, liblsfc), Cint,
                (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Float64, Float64, Ptr{Float64}, Cuint, Cint),
                plan, length(x), length(y), length(z), abs(x[end] - x[1]) + h, k, nuv, flags, device))
    FastMHIP(plan[], nuv, length(x), length(y), length(z), k, quadRule)
end

# The same builder on several GPUs, driven from THIS one Julia process (lsfc_plan_create_gv3d_multi): z-slabs over
# `devices` (0-based HIP device ids), RCCL exchanges over xGMI inside the library.  The returned object is an ordinary
# FastMHIP: `*`, `mul!`, `FFTconvolution`, `gmres_hip!` take and return full host vectors.
function buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nu, devices::Vector{<:Integer}; quadRule::String="Greengard_Vico", flags=0)
    nuv = Vector{Float64}(nu(X, Y, Z)); plan = Ref{Ptr{Cvoid}}(C_NULL); devs = Cint.(devices)
    check(ccall((:lsfc_plan_create_gv3d_multi, liblsfc), Cint,
                (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Float64, Float64, Ptr{Float64}, Cuint, Ptr{Cint}, Cint),
                plan, length(x), length(y), length(z), abs(x[end] - x[1]) + h, k, nuv, flags, devs, length(devs)))
    FastMHIP(plan[], nuv, length(x), length(y), length(z), k, quadRule)
end

# buildFastConvolution(x,y,h,k,nu; quadRule) -- src/FastConvolution.jl:170
function buildFastConvolution(x, y, h, k, nu::Function; quadRule::String="trapezoidal", flags=0, device=0)
    n, m = length(x), length(y)
    X = repeat(x, 1, m)[:]; Y = repeat(y', n, 1)[:]
    nuv = Vector{Float64}(nu(X, Y)); plan = Ref{Ptr{Cvoid}}(C_NULL)
    if quadRule == "trapezoidal"
        D = [1-0.892im, 1-1.35im, 1-1.79im, 1-2.23im, 1-2.67im, 1-3.11im]; D0 = D[round(Int, k*h)]
        check(ccall((:lsfc_plan_create_trap2d, liblsfc), Cint,
                    (Ref{Ptr{Cvoid}}, Int64, Int64, Float64, Float64, Float64, Float64, Float64, Float64, Ptr{Float64}, Cuint, Cint),
                    plan, n, m, x[1], y[1], h, k, real(D0), imag(D0), nuv, flags, device))
    else
        check(ccall((:lsfc_plan_create_gv2d, liblsfc), Cint,
                    (Ref{Ptr{Cvoid}}, Int64, Int64, Float64, Float64, Ptr{Float64}, Cuint, Cint),
                    plan, n, m, abs(x[end] - x[1]) + h, k, nuv, flags, device))
    end
    FastMHIP(plan[], nuv, n, m, 1, k, quadRule)
end

# traits -- src/FastConvolution.jl:31-41
Base.size(M::FastMHIP, dim) = length(M.nu)
Base.size(M::FastMHIP) = (size(M.nu), size(M.nu))
Base.eltype(::FastMHIP) = Complex{Float64}

# fastconvolution / * / mul! -- src/FastConvolution.jl:43-107, src/FastConvolution3D.jl:31-37
function fastconvolution(M::FastMHIP, b::AbstractArray{Complex{Float64},1})
    x = Vector{Complex{Float64}}(b); y = similar(x)
    check(ccall((:lsfc_apply, liblsfc), Cint, (Ptr{Cvoid}, Ptr{Complex{Float64}}, Ptr{Complex{Float64}}, Cint), M.plan, x, y, 0))
    y
end
*(M::FastMHIP, b::AbstractArray{Complex{Float64},1}) = fastconvolution(M, b)
function LinearAlgebra.mul!(Y::AbstractArray{Complex{Float64},1}, M::FastMHIP, b::AbstractArray{Complex{Float64},1})
    Y[:] = M * b
end
# dense vectors (what gmres! hands over): straight into the caller's memory, no temporaries -- at 512^3 the two 2-GB host copies of
# the generic method above cost more than the apply; Y may alias b
function LinearAlgebra.mul!(Y::Vector{Complex{Float64}}, M::FastMHIP, b::Vector{Complex{Float64}})
    check(ccall((:lsfc_apply, liblsfc), Cint, (Ptr{Cvoid}, Ptr{Complex{Float64}}, Ptr{Complex{Float64}}, Cint), M.plan, b, Y, 0))
    Y
end
# page-lock long-lived work vectors of the solver (optional; ~4 ms per 512^3 apply): lsfc_host_register / lsfc_host_unregister
host_register!(v::Vector{Complex{Float64}}) = (check(ccall((:lsfc_host_register, liblsfc), Cint, (Ptr{Cvoid}, Csize_t), v, sizeof(v))); v)
host_unregister!(v::Vector{Complex{Float64}}) = (check(ccall((:lsfc_host_unregister, liblsfc), Cint, (Ptr{Cvoid},), v)); v)
# ... or, preferred, work vectors in page-locked memory owned by the HIP runtime (lsfc_host_alloc): page-aligned, no page shared with the heap
function host_vector(N::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lsfc_host_alloc, liblsfc), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, N * sizeof(Complex{Float64})))
    v = unsafe_wrap(Vector{Complex{Float64}}, Ptr{Complex{Float64}}(p[]), N; own = false)
    finalizer(_ -> ccall((:lsfc_host_free, liblsfc), Cint, (Ptr{Cvoid},), p[]), v)
    v
end

# FFTconvolution -- src/FastConvolution.jl:110-154 (nu only in the 2D trapezoidal branch), src/FastConvolution3D.jl:39-63
function FFTconvolution(M::FastMHIP, b::Array{Complex{Float64},1})
    y = similar(b); apply_nu = (M.l == 1 && M.quadRule == "trapezoidal") ? 1 : 0
    check(ccall((:lsfc_convolve, liblsfc), Cint, (Ptr{Cvoid}, Ptr{Complex{Float64}}, Ptr{Complex{Float64}}, Cint, Cint), M.plan, b, y, apply_nu, 0))
    y
end

# sampleG3D(k,X,Y,Z,indS,fastconv) / sampleGConv -- src/FastConvolution3D.jl:136-160, src/FastConvolution.jl:278-306
# (1-based indS as in the reference; rows of the result are the responses to the delta sources)
function sampleG3D(k, X, Y, Z, indS, M::FastMHIP)
    N = length(M.nu); ns = length(indS)
    out = Array{Complex{Float64}}(undef, N, ns); src = Int64.(indS .- 1)
    check(ccall((:lsfc_sample_sources, liblsfc), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int64, Ptr{Complex{Float64}}, Cint), M.plan, src, ns, out, 0))
    return permutedims(out)          # (ns, N) like the reference's Gc
end
sampleGConv(k, X, Y, indS, M::FastMHIP) = sampleG3D(k, X, Y, nothing, indS, M)

# SparsifyingPreconditioner(Msp, As) with the apply on the device -- src/preconditioner.jl:27-58, 132-170.
# lu(Msp) stays on the host (UMFPACK, as in the reference); its factors go to the device once:
# (F.Rs .* Msp)[F.p, F.q] == F.L * F.U.  CSR arrays of a SparseMatrixCSC X are the CSC arrays of transpose(X).
using SparseArrays
mutable struct SparsifyingPreconditionerHIP
    pc::Ptr{Cvoid}
    N::Int64
end
_csr(X) = (T = sparse(transpose(X)); (Int64.(T.colptr .- 1), Int64.(T.rowval .- 1), Vector{Complex{Float64}}(T.nzval)))
function SparsifyingPreconditionerHIP(Msp::SparseMatrixCSC{Complex{Float64},Int64}, As::SparseMatrixCSC{Complex{Float64},Int64}; device=0)
    F = lu(Msp); N = size(Msp, 1)
    (ap, ac, av) = _csr(As); (lp, lc, lv) = _csr(F.L); (up, uc, uv) = _csr(F.U)
    p = Int64.(F.p .- 1); q = Int64.(F.q .- 1); Rs = Vector{Float64}(F.Rs)
    pc = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lsfc_precond_create, liblsfc), Cint,
                (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Complex{Float64}}, Ptr{Int64}, Ptr{Int64}, Ptr{Complex{Float64}},
                 Ptr{Int64}, Ptr{Int64}, Ptr{Complex{Float64}}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint),
                pc, N, ap, ac, av, lp, lc, lv, up, uc, uv, p, q, Rs, device))
    P = SparsifyingPreconditionerHIP(pc[], N)
    finalizer(P -> ccall((:lsfc_precond_destroy, liblsfc), Cint, (Ptr{Cvoid},), P.pc), P)
    return P
end
# ldiv!(P, b): host vector in, host vector out (staged over PCIe); inside gmres_hip! the device path is used instead
function LinearAlgebra.ldiv!(P::SparsifyingPreconditionerHIP, b::Vector{Complex{Float64}})
    check(ccall((:lsfc_precond_apply, liblsfc), Cint, (Ptr{Cvoid}, Ptr{Complex{Float64}}, Cint), P.pc, b, 0)); b
end
Base.:\(P::SparsifyingPreconditionerHIP, b::Vector{Complex{Float64}}) = ldiv!(P, copy(b))

# Device-side GMRES with a host preconditioner: Pl is anything with the two-argument ldiv!(Pl, v)
# (src/preconditioner.jl:147-170), passed through @cfunction.
# C layout of lsfc_gmres_opts / lsfc_gmres_result (include/lsfc.h), field: byte offset -- tests/test_abi.py compiles the same
# table as static_asserts against the header and checks that this comment agrees with it (Julia lays isbits structs out like C)
# ABI-LAYOUT lsfc_gmres_opts size=64 restart:0 maxiter:8 reltol:16 abstol:24 orth:32 initially_zero:36 precond:40 precond_user:48 precond_on_device:56
# ABI-LAYOUT lsfc_gmres_result size=32 iters:0 mvps:8 converged:16 final_resnorm:24
struct GmresOpts
    restart::Cint; maxiter::Int64; reltol::Float64; abstol::Float64; orth::Cint; initially_zero::Cint
    precond::Ptr{Cvoid}; precond_user::Ptr{Cvoid}; precond_on_device::Cint
end
struct GmresResult
    iters::Int64; mvps::Int64; converged::Cint; final_resnorm::Float64
end
function _precond_trampoline(user::Ptr{Cvoid}, v::Ptr{Float64}, n::Int64)::Cint
    Pl = unsafe_pointer_to_objref(user)[]
    ldiv!(Pl, unsafe_wrap(Array, Ptr{Complex{Float64}}(v), n))
    return Cint(0)
end
function gmres_hip!(x::Vector{Complex{Float64}}, M::FastMHIP, b::Vector{Complex{Float64}}; Pl=nothing, restart=min(20, length(b)),
                    maxiter=length(b), reltol=sqrt(eps(Float64)), abstol=0.0, initially_zero=false)
    box = Ref{Any}(Pl)
    if Pl isa SparsifyingPreconditionerHIP        # applied on the device by the library itself: no PCIe, no Julia in the loop
        cb = cglobal((:lsfc_precond_callback, liblsfc)); user = Pl.pc; ondev = Cint(1)
    else
        cb = Pl === nothing ? C_NULL : @cfunction(_precond_trampoline, Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64))
        user = Pl === nothing ? C_NULL : pointer_from_objref(box); ondev = Cint(0)
    end
    opts = Ref(GmresOpts(restart, maxiter, reltol, abstol, 0, initially_zero ? 1 : 0, cb, user, ondev))
    res = Ref(GmresResult(0, 0, 0, 0.0)); resnorm = zeros(Float64, maxiter)
    GC.@preserve box begin
        rc = ccall((:lsfc_gmres, liblsfc), Cint,
                   (Ptr{Cvoid}, Ptr{Complex{Float64}}, Ptr{Complex{Float64}}, Ref{GmresOpts}, Ptr{Float64}, Int64, Ref{GmresResult}, Cint),
                   M.plan, x, b, opts, resnorm, maxiter, res, 0)
        (rc == 0 || rc == -5) || check(rc)
    end
    x, resnorm[1:res[].iters]
end

end # module
