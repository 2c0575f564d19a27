function [p,v] = propStatedmpc(po, vo, a, A_initp, A_p, A_v)
% Shadows dmpc/matlab/propStatedmpc.m (same signature): p = A_p a + A_initp [po;vo], v = A_v a + repmat(vo), on the GPU.
K = length(a)/3;
prm = dmpc_params_struct(0, 0.2, K, 0.35, [-1 -1 0], [1 1 1], 1, 1000, 100, eye(3), 2, -5e4);   % context only
[p,v] = dmpc_mex('prop_state', prm, A_p, A_v, A_initp, po(:), vo(:), a(:));
end
