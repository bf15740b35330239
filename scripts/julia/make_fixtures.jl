# make_fixtures.jl -- golden vectors of the contact hot path from the REFERENCE ITSELF (PressureFieldContact.jl).
#
# Nobody could run this in the build container (no Julia toolchain there); it is committed so that a maintainer who
# has Julia >= 1.1 with the reference's Manifest can close the last parity gap (SURVEY.md §8(c), VERDICT round 1):
#
#     julia --project=/path/to/PressureFieldContact.jl scripts/julia/make_fixtures.jl tests/golden/julia_fixtures.txt
#     python -m pytest tests/test_julia_fixtures.py          # oracle (CPU); add -m gpu on an MI355X box for the HIP path
#
# For every contact instruction of a few small scenes it runs the reference's own force_single_elastic_intersection!
# pieces (src/contact_algorithms_non_friction.jl:70-84,94-143) and dumps, in a flat text format,
#   inputs : both eMeshes, both bin_BB_Tree{OBB} flattened in preorder (the host-supplied-tree path of pfc_add_mesh:
#            the builder's tie-breaking depends on Julia Dict order, so the tree travels with the fixture), the
#            instruction parameters, x_r2_r1 / x_r1_r2, twist_r2_r1_r2, the bristle state s
#   outputs: TT_Cache.vc (candidate pairs, 0-based), length(poly_ζ²) per pair, the TractionCache, the wrench on body 2
#            (as_static_vector), ṡ, and for bristle instructions K, K̄^{-1/2}, S⁻¹.
# Record layout: `key n` on one line, then n whitespace-separated numbers on the next (Float64 printed with 17
# significant digits, i.e. round-trip exact).  tests/test_julia_fixtures.py reads it.
using StaticArrays, LinearAlgebra, Printf
using RigidBodyDynamics
using PressureFieldContact
const PFC = PressureFieldContact
using PressureFieldContact.Binary_BB_Trees: bin_BB_Tree, OBB, is_leaf, tree_tree_intersect, update_TT_Cache!
using PressureFieldContact.Clip: clip_in_tet_coordinates, clip_plane_tet, one_pad_then_mul, zero_small_coordinates

set_zero_subnormals(true)
BLAS.set_num_threads(1)

put(io, key, v::AbstractVector{<:Integer}) = (println(io, key, " ", length(v)); println(io, join(string.(v), " ")))
put(io, key, v::AbstractVector{<:Real}) = (println(io, key, " ", length(v)); println(io, join([@sprintf("%.17g", Float64(x)) for x in v], " ")))
put(io, key, x::Real) = put(io, key, [x])
flat(v::Vector{<:SVector}) = collect(Iterators.flatten(v))

# preorder flattening of bin_BB_Tree{OBB}: node 0 = root; children as 0-based node indices (-1 for leaves); leaf id
# 0-based element index (the reference's -9999 sentinel marks internal nodes, src/obb/tree_types.jl:11,56)
function flatten_tree(tree::bin_BB_Tree)
    c = Float64[]; e = Float64[]; R = Float64[]; child = Int[]; leaf = Int[]
    function visit(n)
        k = length(leaf)
        append!(c, n.box.c); append!(e, n.box.e); append!(R, vec(Matrix(n.box.R)))   # column-major
        push!(leaf, is_leaf(n) ? n.id - 1 : -9999)
        push!(child, -1); push!(child, -1)
        if !is_leaf(n)
            child[2k + 1] = visit(n.node_1)
            child[2k + 2] = visit(n.node_2)
        end
        return k
    end
    visit(tree)
    return c, e, R, child, leaf
end

function put_mesh(io, tag, mc)
    put(io, "$(tag)_point", flat(mc.mesh.point))
    (mc.mesh.tri === nothing) || put(io, "$(tag)_tri", flat(mc.mesh.tri) .- 1)
    (mc.mesh.tet === nothing) || (put(io, "$(tag)_tet", flat(mc.mesh.tet) .- 1); put(io, "$(tag)_eps", mc.mesh.ϵ))
    put(io, "$(tag)_Ebar", mc.c_prop === nothing ? 0.0 : mc.c_prop.Ē)
    c, e, R, child, leaf = flatten_tree(mc.tree)
    put(io, "$(tag)_node_c", c); put(io, "$(tag)_node_e", e); put(io, "$(tag)_node_R", R)
    put(io, "$(tag)_node_child", child); put(io, "$(tag)_node_leaf", leaf)
end

# length(poly_ζ²) for one candidate pair, with the reference's own clip calls (non_friction.jl:166-215)
function clip_count(i_1, i_2, mesh_1, mesh_2, b)
    vert_2, ϵ² = PFC.tetrahedron_vertices_ϵ(i_2, mesh_2)
    x_r²_ζ², x_ζ²_r² = PFC.calc_ζ_transforms(vert_2)
    if mesh_1.mesh.tri !== nothing
        vert_1 = PFC.triangle_vertices(i_1, mesh_1)
        x_ζ²_r¹ = x_ζ²_r² * b.x_r²_r¹.mat
        v = [x_ζ²_r¹ * PFC.onePad(vert_1[k]) for k = 1:3]
        return length(clip_in_tet_coordinates(v[1], v[2], v[3]))
    else
        vert_1, ϵ¹ = PFC.tetrahedron_vertices_ϵ(i_1, mesh_1)
        x_r¹_ζ¹, x_ζ¹_r¹ = PFC.calc_ζ_transforms(vert_1)
        ϵ_plane_1_r² = PFC.find_plane_tet(PFC.get_Ē(mesh_1), ϵ¹, x_ζ¹_r¹ * b.x_r¹_r².mat)
        ϵ_plane_2_r² = PFC.find_plane_tet(PFC.get_Ē(mesh_2), ϵ², x_ζ²_r²)
        ϵ_plane_r² = ϵ_plane_2_r² - ϵ_plane_1_r²
        poly_r² = clip_plane_tet(ϵ_plane_r², b.x_r²_r¹.mat * x_r¹_ζ¹)
        (3 <= length(poly_r²)) || return 0
        poly_ζ² = zero_small_coordinates(one_pad_then_mul(x_ζ²_r², poly_r²))
        return length(clip_in_tet_coordinates(poly_ζ²))
    end
end

function dump_scene(io, name::String, m)
    x = get_state(m)
    calcXd(x, m)                    # the reference's own evaluation (also warms every cache)
    tm = m.float
    for (k, c_ins) in enumerate(m.ContactInstructions)
        println(io, "case ", name, "_ins", k)
        # --- exactly force_single_elastic_intersection! (:70-84), one instruction at a time
        PFC.calcTriTetIntersections!(m, c_ins)
        b = tm.bodyBodyCache
        PFC.refreshBodyBodyCache!(m, tm, c_ins)
        put_mesh(io, "m1", b.mesh_1); put_mesh(io, "m2", b.mesh_2)
        fm = c_ins.FrictionModel
        is_bristle = isa(fm, PFC.Bristle)
        n_quad = length(c_ins.quad.w) == 1 ? 1 : 2
        put(io, "chi", c_ins.χ); put(io, "n_quad", [n_quad]); put(io, "model", [is_bristle ? 1 : 0])
        put(io, "mu_s", fm.μs); put(io, "mu_d", fm.μd)
        is_bristle ? (put(io, "tau", fm.τ); put(io, "k_bar", fm.k̄); put(io, "magic", fm.magic)) : put(io, "v_c", fm.v_c)
        R21 = rotation(b.x_r²_r¹); t21 = translation(b.x_r²_r¹); R12 = rotation(b.x_r¹_r²); t12 = translation(b.x_r¹_r²)
        put(io, "pose", vcat(vec(Matrix(R21)), t21, vec(Matrix(R12)), t12))
        put(io, "twist", PFC.as_static_vector(b.twist_r²_r¹_r²))
        s = is_bristle ? collect(PFC.get_bristle_d0(tm, fm.BristleID)) : zeros(6)
        put(io, "s", s)
        n_pair = length(m.TT_Cache.vc)
        pairs = Int[]; clip_n = Int[]
        for j = 1:n_pair
            i_1, i_2 = m.TT_Cache.vc[j]
            push!(pairs, i_1 - 1); push!(pairs, i_2 - 1)
            push!(clip_n, clip_count(i_1, i_2, b.mesh_1, b.mesh_2, b))
        end
        put(io, "pairs", pairs); put(io, "clip_n", clip_n)
        wrench = zeros(6); sdot = zeros(6)
        if n_pair != 0
            PFC.integrate_over!(b, m.TT_Cache)
        end
        tc = b.TractionCache
        trac = Float64[]
        for j = 1:length(tc)
            t = tc[j]
            append!(trac, t.n̂); append!(trac, t.r_cart); push!(trac, t.dA); push!(trac, t.p)
        end
        put(io, "trac", trac)
        if n_pair != 0 && !isempty(tc)
            w = PFC.yes_contact!(fm, tm, c_ins)
            wrench = collect(PFC.as_static_vector(w))
            if is_bristle
                ss = b.spatialStiffness
                put(io, "K", vec(Matrix(ss.K))); put(io, "Kbar_inv_sqrt", vec(Matrix(ss.K̄⁻¹_sqrt))); put(io, "Sinv", collect(ss.S⁻¹.diag))
            end
        else
            PFC.no_contact!(fm, tm, c_ins)
        end
        is_bristle && (sdot = collect(PFC.get_bristle_d1(tm, fm.BristleID)))
        put(io, "wrench", wrench); put(io, "sdot", sdot)
        println(io, "end")
    end
end

# ---- scenes -----------------------------------------------------------------------------------------------------------
function scene_boxes()      # test/boxes.jl:18-45 geometry; boxes stacked 1 mm into each other so that every instruction is in contact
    box_rad = 0.05
    c_prop = ContactProperties(Ē=1.0e6)
    i_c = InertiaProperties(400.0); i_r = InertiaProperties(400.0, d=box_rad)
    eM_r = as_tri_eMesh(eMesh_box(box_rad)); eM_c = as_tet_eMesh(eMesh_box(box_rad))
    m = MechanismScenario()
    nt_plane = add_contact!(m, "plane", as_tet_eMesh(eMesh_half_plane()), c_prop=c_prop)
    b1 = add_body_contact!(m, "box_1", eM_r, i_prop=i_r)
    b2 = add_body_contact!(m, "box_2", eM_c, i_prop=i_c, c_prop=c_prop)
    b3 = add_body_contact!(m, "box_3", eM_r, i_prop=i_r)
    b4 = add_body_contact!(m, "box_4", eM_c, i_prop=i_c, c_prop=c_prop)
    add_friction_regularize!(m, nt_plane.id, b1.id, μd=0.0, χ=2.2, n_quad_rule=2)
    add_friction_regularize!(m, b1.id, b2.id, μd=0.2, χ=0.2, n_quad_rule=2)
    add_friction_regularize!(m, b2.id, b3.id, μd=0.2, χ=0.2, n_quad_rule=2)
    add_friction_regularize!(m, b3.id, b4.id, μd=0.2, χ=0.2, n_quad_rule=2)
    finalize!(m)
    pen = 0.001
    for (k, b) in enumerate((b1, b2, b3, b4))
        set_state_spq!(m, b.joint, trans=SVector(0.013 * k, -0.007 * k, (2k - 1) * box_rad - k * pen),
                       rot=RotZ(0.1 * k) * RotX(0.02 * k), w=SVector(0.1, -0.2, 1.0 * k), vel=SVector(0.01 * k, 0.02, -0.03))
    end
    return m
end

function scene_bristle(k_quad::Int)   # test/test_normal.jl:2-25 geometry with a tilted, moving box and a non-zero bristle state
    box_rad = 0.05
    m = MechanismScenario()
    add_contact!(m, "plane", as_tet_eMesh(eMesh_half_plane()), c_prop=ContactProperties(Ē=1.0e9))
    eM_box = as_tri_eMesh(eMesh_box(box_rad))
    transform!(eM_box, SVector{3,Float64}(0.0, 0.0, box_rad))
    body, joint = add_body_contact!(m, "box", eM_box, i_prop=InertiaProperties(400.0, d=0.09))
    add_friction_bristle!(m, find_mesh_id(m, "box"), find_mesh_id(m, "plane"), μd=0.3, χ=0.6, k̄=1.0e6, τ=0.03, n_quad_rule=k_quad)
    finalize!(m)
    set_state_spq!(m, joint, trans=SVector(0.1, 0.2, -0.1 * box_rad), rot=RotZ(0.3) * RotY(0.01),
                   w=SVector(0.2, -0.1, 0.5), vel=SVector(0.03, -0.02, -0.01))
    m.float.s .= [1.0e-3, -2.0e-3, 0.5e-3, 2.0e-4, -1.0e-4, 3.0e-4]
    return m
end

function scene_vol_vol()    # test/test_vol_vol.jl geometry: compliant box on the compliant half-plane (tet-tet)
    box_rad = 0.05
    c_prop = ContactProperties(Ē=1.0e6)
    m = MechanismScenario()
    add_contact!(m, "plane", as_tet_eMesh(eMesh_half_plane()), c_prop=c_prop)
    body, joint = add_body_contact!(m, "box", as_tet_eMesh(eMesh_box(box_rad)), i_prop=InertiaProperties(400.0), c_prop=c_prop)
    add_friction_regularize!(m, find_mesh_id(m, "plane"), find_mesh_id(m, "box"), μd=0.3)
    finalize!(m)
    set_state_spq!(m, joint, trans=SVector(0.02, -0.03, box_rad - 0.002), rot=RotZ(0.4), w=SVector(0.0, 0.1, 1.0), vel=SVector(0.05, 0.0, -0.02))
    return m
end

out = length(ARGS) >= 1 ? ARGS[1] : "julia_fixtures.txt"
open(out, "w") do io
    println(io, "# PressureFieldContact.jl reference fixtures; generated by scripts/julia/make_fixtures.jl")
    dump_scene(io, "boxes", scene_boxes())
    dump_scene(io, "bristle_q1", scene_bristle(1))
    dump_scene(io, "bristle_q2", scene_bristle(2))
    dump_scene(io, "vol_vol", scene_vol_vol())
end
println("wrote ", out)
