function [po,pf] = randomExchange(N,pmin,pmax,rmin)
% Shadows dmpc/matlab/randomExchange.m (same signature): starts by rejection sampling (Euclidean separation), goals = the
% starts permuted so that no agent keeps its own, generated on the GPU (own counter-based random stream).
prm = dmpc_params_struct(0, 0.2, 15, max(rmin,0.01), pmin, pmax, 1, 1000, 100, eye(3), 2, -5e4);   % context only
[a,b] = dmpc_mex('random_exchange', prm, N, pmin(:)', pmax(:)', rmin, floor(rand*2^52));
po = reshape(a,1,3,N); pf = reshape(b,1,3,N);
end
