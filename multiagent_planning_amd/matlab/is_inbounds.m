function inbounds = is_inbounds(p,pmin,pmax)
% Shadows dmpc/matlab/is_inbounds.m (same signature): every column of p (3 x n) inside [pmin - 5 cm, pmax + 5 cm], on the GPU.
prm = dmpc_params_struct(0, 0.2, 15, 0.35, pmin, pmax, 1, 1000, 100, eye(3), 2, -5e4);   % context only
inbounds = dmpc_mex('is_inbounds', prm, reshape(p,3,[]), pmin(:), pmax(:));
end
