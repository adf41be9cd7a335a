# BarBayHIP.jl -- reference-side binding of libbarbay_hip.so (include/barbay_hip.h).
#
# UNTESTED IN THIS REPOSITORY'S IMAGE (no `julia`); written against Julia 1.x `ccall` semantics and kept
# line-for-line with the Python binding barbay.jl_amd/_capi.py, which IS tested.  It replaces exactly
#     q = Turing.vi(bayes_model, advi; optimizer=opt)                      (BarBay.jl src/vi.jl:201)
# and returns an object with the three fields `BarBay.utils.advi_to_df` reads
# (`q.dist.m`, `q.dist.σ`, `q.transform.ranges_out`; src/utils.jl:1049, 1060).
module BarBayHIP

const LIB = get(ENV, "BARBAY_HIP_LIB", joinpath(@__DIR__, "..", "barbay.jl_amd", "lib", "libbarbay_hip.so"))

struct bb_prior
    mean::Ptr{Float64}
    std::Ptr{Float64}
    n::Int64
end
bb_prior() = bb_prior(C_NULL, C_NULL, 0)

struct bb_model_desc
    kind::Int32
    n_rep::Int32
    n_neutral::Int64
    n_bc::Int64
    n_time::Ptr{Int32}
    counts::Ptr{Int64}
    totals::Ptr{Int64}
    n_env::Int32
    env_idx::Ptr{Int32}
    n_geno::Int32
    geno_idx::Ptr{Int32}
    s_pop_prior::bb_prior
    logsigma_pop_prior::bb_prior
    s_bc_prior::bb_prior
    logsigma_bc_prior::bb_prior
    loglambda_prior::bb_prior
    logtau_prior::bb_prior
    flags::Int32
end

mutable struct bb_advi_opts
    samples_per_step::Int32
    optimizer::Int32
    eta::Float64
    tau::Float64
    window::Int32
    resum_every::Int32
    pre::Float64
    post::Float64
    seed::UInt64
    device::Int32
    rank::Int32
    world_size::Int32
    steps_per_graph::Int32
    elbo_every::Int32
    launch_mode::Int32
    n_devices::Int32
    device_ids::Ptr{Int32}
    bb_advi_opts() = new()
end

struct bb_block_range
    name::NTuple{24,UInt8}
    lo::Int64
    hi::Int64
end

check(rc) = rc == 0 || error("barbay_hip: " * unsafe_string(ccall((:bb_last_error, LIB), Cstring, ())))

# what advi_to_df needs from `q`
struct Dist; m::Vector{Float64}; σ::Vector{Float64}; end
struct Transform; ranges_out::Vector{UnitRange{Int}}; end
struct Posterior
    dist::Dist
    transform::Transform
    hier::Union{Nothing,NamedTuple}     # device-side `process_hierarchical_samples!` result (median, std per θ̃ unit), if any
end

const KIND = Dict("fitness_normal" => 0, "multienv_fitness_normal" => 1, "genotype_fitness_normal" => 2,
                  "replicate_fitness_normal" => 3, "multienv_replicate_fitness_normal" => 4)

"""
    group_genotypes(data; genotype_col=:genotype, neutral_col=:neutral) -> data

Stable reorder of a tidy frame's rows so that the mutant barcodes of one genotype are consecutive (genotypes in order of first
appearance).  NOT needed any more: `bb_create` groups the mutants itself where `geno_idx` is not in consecutive runs and presents
the caller's order at the ABI (`bb_get_permutation` tells the mapping); kept as a convenience.  Results are
keyed by barcode id, so the order is the caller's to choose.  (`data_to_arrays` keeps barcodes in order of appearance,
src/utils.jl:692-731.)  Works on any Tables.jl-style object with `getproperty` columns and `data[perm, :]` indexing (DataFrame).
"""
function group_genotypes(data; genotype_col::Symbol=:genotype, neutral_col::Symbol=:neutral)
    g = getproperty(data, genotype_col)
    neutral = getproperty(data, neutral_col)
    first_seen = Dict{eltype(g),Int}()
    for x in g
        get!(first_seen, x, length(first_seen) + 1)
    end
    key = [neutral[i] ? 0 : first_seen[g[i]] for i in eachindex(g)]        # neutrals keep their place in front
    return data[sortperm(key; alg=MergeSort), :]
end

# prior kwarg (`VecOrMat{Float64}`) -> (mean, std) vectors kept alive by the caller
_prior_arrays(p::Vector{Float64}) = ([p[1]], [p[2]])
_prior_arrays(p::Matrix{Float64}) = (p[:, 1], p[:, 2])

"""
    vi(model_name, R, n_t, n_neutral, n_bc; samples_per_step, max_iters, optimizer, priors..., envs, genotypes, seed)

Drop-in for `Turing.vi(bayes_model, advi; optimizer=opt)`.  `devices = collect(0:7)` runs the one call on all GPUs of the node.  `R` / `n_t` are `data_arrays.bc_count` /
`data_arrays.bc_total` exactly as `utils.data_to_arrays` returns them (Matrix, 3-D Array or Vector{Matrix}).
"""
function vi(model_name::String, R, n_t, n_neutral::Int, n_bc::Int;
            samples_per_step::Int=1, max_iters::Int=10_000,
            optimizer::Symbol=:TruncatedADAGrad, eta=0.1, tau=40.0, n=100, pre=1.0, post=0.9,
            priors::Dict{Symbol,<:Any}=Dict{Symbol,Any}(), envs=nothing, genotypes=nothing,
            seed::Integer=0, device::Integer=0, devices::Vector{<:Integer}=Int[], hier_samples::Integer=10_000, verbose::Bool=false)
    mats = R isa Vector ? R : (ndims(R) == 3 ? [R[:, :, r] for r in axes(R, 3)] : [R])
    tots = n_t isa Vector{<:Vector} ? n_t : (ndims(n_t) == 2 ? [n_t[:, r] for r in axes(n_t, 2)] : [n_t])
    n_time = Int32[size(m, 1) for m in mats]
    counts = reduce(vcat, vec.(mats))              # column-major T x B, t fastest: passed as is
    totals = reduce(vcat, tots)
    # env list per replicate, replicate-major (the 3-D multienv_replicate method repeats its one list per replicate)
    envs_rep = envs === nothing ? nothing : (envs isa Vector{<:Vector} ? envs : [envs for _ in mats])
    env_flat = envs_rep === nothing ? nothing : reduce(vcat, envs_rep)
    env_idx = env_flat === nothing ? Int32[] : Int32.(indexin(env_flat, unique(env_flat)) .- 1)
    geno_idx = genotypes === nothing ? Int32[] : Int32.(indexin(genotypes, unique(genotypes)) .- 1)
    pa = Dict(k => _prior_arrays(v) for (k, v) in priors)
    pr(k) = haskey(pa, k) ? bb_prior(pointer(pa[k][1]), pointer(pa[k][2]), length(pa[k][1])) : bb_prior()
    opts = bb_advi_opts()
    ccall((:bb_default_opts, LIB), Cvoid, (Ref{bb_advi_opts},), opts)
    opts.samples_per_step = samples_per_step
    opts.optimizer = optimizer == :TruncatedADAGrad ? 0 : 1
    opts.eta, opts.tau, opts.window, opts.pre, opts.post = eta, tau, n, pre, post
    opts.seed, opts.device = seed, device
    # devices = [0, 1, ..., 7]: this one call drives all of them (the barcodes shard over the devices inside the library,
    # resident launches exchanging over xGMI); empty or one entry: a single GPU
    dev_ids = Int32.(devices)
    if length(dev_ids) > 1
        opts.n_devices, opts.device_ids = length(dev_ids), pointer(dev_ids)
    elseif length(dev_ids) == 1
        opts.device = dev_ids[1]
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve n_time counts totals env_idx geno_idx pa dev_ids begin
        md = bb_model_desc(KIND[model_name], length(mats), n_neutral, n_bc, pointer(n_time), pointer(counts),
                           pointer(totals), isempty(env_idx) ? 0 : maximum(env_idx) + 1,
                           isempty(env_idx) ? C_NULL : pointer(env_idx),
                           isempty(geno_idx) ? 0 : maximum(geno_idx) + 1,
                           isempty(geno_idx) ? C_NULL : pointer(geno_idx),
                           pr(:s_pop_prior), pr(:logσ_pop_prior), pr(:s_bc_prior), pr(:logσ_bc_prior),
                           pr(:logλ_prior), pr(:logτ_prior),
                           Int32(R isa Vector ? 1 : 0))   # BB_FLAG_RAGGED_METHOD: the Vector{Matrix} method was dispatched
        check(ccall((:bb_create, LIB), Cint, (Ref{bb_model_desc}, Ref{bb_advi_opts}, Ref{Ptr{Cvoid}}), md, opts, h))
    end
    try
        if verbose      # (`BarBay.vi.advi(...; verbose)`, src/vi.jl:122-124): which kernel the run launches -- bb_kernel_name
            nm = Vector{UInt8}(undef, 128)
            check(ccall((:bb_kernel_name, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int64), h[], nm, 128))
            @info "BarBayHIP: " * unsafe_string(pointer(nm))
        end
        check(ccall((:bb_run, LIB), Cint, (Ptr{Cvoid}, Int64), h[], max_iters))
        D = ccall((:bb_num_latents, LIB), Int64, (Ptr{Cvoid},), h[])
        m, s = Vector{Float64}(undef, D), Vector{Float64}(undef, D)
        check(ccall((:bb_get_posterior, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], m, s))
        blocks = Vector{bb_block_range}(undef, 8)
        nb = Ref{Int32}(0)
        check(ccall((:bb_get_layout, LIB), Cint, (Ptr{Cvoid}, Ptr{bb_block_range}, Ref{Int32}), h[], blocks, nb))
        ranges = [Int(b.lo)+1:Int(b.hi) for b in blocks[1:nb[]]]
        hier = nothing
        nu = ccall((:bb_hier_units, LIB), Int64, (Ptr{Cvoid},), h[])
        if nu > 0 && hier_samples > 0          # src/utils.jl:1284-1343 on the device: 10 000 draws per unit, median + std
            med, sd = Vector{Float64}(undef, nu), Vector{Float64}(undef, nu)
            check(ccall((:bb_hier_fitness, LIB), Cint, (Ptr{Cvoid}, Int32, UInt64, Ptr{Float64}, Ptr{Float64}),
                        h[], hier_samples, seed, med, sd))
            hier = (n_samples=hier_samples, median=med, std=sd)
        end
        return Posterior(Dist(m, s), Transform(ranges), hier)
    finally
        ccall((:bb_destroy, LIB), Cvoid, (Ptr{Cvoid},), h[])
    end
end

end # module
