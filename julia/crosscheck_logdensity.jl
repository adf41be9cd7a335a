# crosscheck_logdensity.jl -- OPPORTUNISTIC pin of the oracle against the real reference (SURVEY.md §8c(iii)).
#
# NOT run anywhere in this repository's pipeline: the build image has no `julia`.  Whoever has Julia with
# BarBay.jl (mrazomej/BarBay.jl @ 2025-01-17) and Turing 0.36 / DynamicPPL 0.32 / LogDensityProblems installed can run
#
#     julia --project=<BarBay.jl checkout> julia/crosscheck_logdensity.jl <path to this repo>
#
# It evaluates the REAL Turing model's log-joint density (and its ReverseDiff gradient) at the points z committed in
# tests/golden/crosscheck_points.json -- the same z the golden fixtures golden_data00*.npz hold -- and prints the
# difference to the value the literal oracle (oracle/literal.py) produced there.  Agreement to ~1e-10 relative pins the
# oracle, and with it every parity test of the HIP engine, against the reference itself; a disagreement names the model
# whose transcription (or whose [recalled] upstream formula) is off.
#
# Latent order: DynamicPPL's flattened VarInfo order == the model's `~` statements in source order, Julia column-major --
# the flat layout of include/barbay_hip.h (bb_get_layout).  All latents are unconstrained Normals: link / invlink are identities.
import BarBay, CSV, DataFrames, JSON, Turing, DynamicPPL, LogDensityProblems, LogDensityProblemsAD, ReverseDiff

repo = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..")
points = JSON.parsefile(joinpath(repo, "tests", "golden", "crosscheck_points.json"))

function build(name)
    df = CSV.read(joinpath(repo, "tests", "golden", name * ".csv"), DataFrames.DataFrame)
    if name == "data001_single"
        d = BarBay.utils.data_to_arrays(df)
        return BarBay.model.fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc)
    elseif name == "data002_hier-rep"
        d = BarBay.utils.data_to_arrays(df; rep_col=:rep)
        return BarBay.model.replicate_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc)
    elseif name == "data003_multienv"
        d = BarBay.utils.data_to_arrays(df; env_col=:env)
        return BarBay.model.multienv_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc; envs=d.envs)
    else
        d = BarBay.utils.data_to_arrays(df; genotype_col=:genotype)
        return BarBay.model.genotype_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc; genotypes=d.genotypes)
    end
end

worst = 0.0
for name in ("data001_single", "data002_hier-rep", "data003_multienv", "data004_multigen")
    p = points[name]
    model = build(name)
    ldf = DynamicPPL.LogDensityFunction(model)          # log p(data, z): priors + likelihood, no Jacobian (identity bijectors)
    z = Float64.(p["z"])
    @assert LogDensityProblems.dimension(ldf) == p["D"] "latent count differs for $name: $(LogDensityProblems.dimension(ldf)) vs $(p["D"])"
    lp = LogDensityProblems.logdensity(ldf, z)
    adf = LogDensityProblemsAD.ADgradient(:ReverseDiff, ldf)
    _, g = LogDensityProblems.logdensity_and_gradient(adf, z)
    rel = abs(lp - p["logjoint"]) / abs(p["logjoint"])
    gerr = maximum(abs.(g[1:8] .- Float64.(p["grad_z_first8"]))) / maximum(abs.(g))
    println(rpad(name, 20), " logjoint reference ", lp, "  oracle ", p["logjoint"], "  rel.diff ", rel, "  grad[1:8] rel.diff ", gerr)
    global worst = max(worst, rel, gerr)
end
println(worst <= 1e-9 ? "ORACLE PINNED: agreement <= 1e-9 on all four fixtures" : "MISMATCH: worst relative difference $worst")
